cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt3; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt3 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-prof > /tmp/kt3.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt3 -name "*.db" | head -1) --md --by-grid > $GRAFT_REPO_ROOT/gpurun_out/r10_stats_cosched.md 2>&1
rm -rf /tmp/kt4; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt4 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-prof --lib-option two_streams=0 > /tmp/kt4.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt4 -name "*.db" | head -1) --md --by-grid > $GRAFT_REPO_ROOT/gpurun_out/r10_stats_serial.md 2>&1
tail -2 /tmp/kt3.log /tmp/kt4.log
head -16 $GRAFT_REPO_ROOT/gpurun_out/r10_stats_cosched.md | cut -c1-150
head -16 $GRAFT_REPO_ROOT/gpurun_out/r10_stats_serial.md | cut -c1-150
