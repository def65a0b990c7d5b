# SAC configs[3]: timeline of ~1.5 vector steps in the middle of the run (kernel, duration, queue)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kts; timeout 300 rocprofv3 --kernel-trace -d /tmp/kts -- python $GRAFT_REPO_ROOT/tools/sac_bench.py full_jit > /tmp/kts.log 2>&1
DB=$(find /tmp/kts -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB 0.6 150 > $GRAFT_REPO_ROOT/gpurun_out/sac_fulljit_timeline.txt 2>&1
tail -150 $GRAFT_REPO_ROOT/gpurun_out/sac_fulljit_timeline.txt
