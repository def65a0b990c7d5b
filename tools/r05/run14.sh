cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt4; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt4 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-prof --lib-option two_streams=0 > /tmp/kt4.log 2>&1
(echo "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-prof --lib-option two_streams=0   (policy and critic chains SERIALISED on one stream: every kernel alone on the chip)"; echo; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt4 -name "*.db" | head -1) --md --by-grid) > $GRAFT_REPO_ROOT/gpurun_out/r05_bench_kernel_stats_serialized_nets.md 2>&1
rm -rf /tmp/ktp; timeout 300 rocprofv3 --kernel-trace -d /tmp/ktp -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary --no-prof > /tmp/ktp.log 2>&1
DB=$(find /tmp/ktp -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB 0.55 70 > $GRAFT_REPO_ROOT/gpurun_out/r05_ppo_timeline.txt 2>&1
rm -rf /tmp/ktq; timeout 300 rocprofv3 --kernel-trace -d /tmp/ktq -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-prof --minibatch-size-global 4096 > /tmp/ktq.log 2>&1
DB=$(find /tmp/ktq -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB 0.55 40 > $GRAFT_REPO_ROOT/gpurun_out/r05_ppo_timeline_mb4096.txt 2>&1
head -20 $GRAFT_REPO_ROOT/gpurun_out/r05_ppo_timeline_mb4096.txt
