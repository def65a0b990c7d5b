# per-rank work of the literal 8-GPU configuration (1280 updates of 4096 rows) on one GPU: timing, kernel stats, timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --steps 3 --warmup 1 --minibatch-size-global 4096 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktm; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktm -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --minibatch-size-global 4096 --no-cpu-baseline --no-secondary > /tmp/ktm.log 2>&1
DB=$(find /tmp/ktm -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB --md > $GRAFT_REPO_ROOT/gpurun_out/smallmb_kernel_stats.md 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB 0.85 120 > $GRAFT_REPO_ROOT/gpurun_out/smallmb_timeline.txt 2>&1
head -32 $GRAFT_REPO_ROOT/gpurun_out/smallmb_kernel_stats.md
