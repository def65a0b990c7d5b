# fastsac.hip at the reference's default sizes: plain timing + rocprofv3 kernel stats of tools/fastsac_bench.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/fastsac_bench.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kts; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kts -- python $GRAFT_REPO_ROOT/tools/fastsac_bench.py > /tmp/kts.log 2>&1 < /dev/null
DB=$(find /tmp/kts -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB --md > $GRAFT_REPO_ROOT/gpurun_out/fastsac_kernel_stats.md 2>&1
  head -45 $GRAFT_REPO_ROOT/gpurun_out/fastsac_kernel_stats.md | cut -c1-200
else tail -5 /tmp/kts.log; fi
