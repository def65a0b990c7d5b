# PPO configs[1]: timeline of ~3 updates in the middle of the iteration (kernel, duration, queue): where the two chains overlap
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktp; timeout 300 rocprofv3 --kernel-trace -d /tmp/ktp -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-prof > /tmp/ktp.log 2>&1
DB=$(find /tmp/ktp -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB 0.75 90 > $GRAFT_REPO_ROOT/gpurun_out/ppo_timeline.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/ppo_timeline.txt
