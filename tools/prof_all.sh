# Round artifacts (run on the GPU box from the repo root): the default bench line, the rocprofv3 kernel table of the same
# workload (per kernel AND launch grid, so the L2 / L3 shapes of one kernel are separate rows), HBM traffic and MFMA counters.
#   bash tools/prof_all.sh [TAG]      -> gpurun_out/${TAG}_bench_n1.json, _bench_kernel_stats.md, _pmc_traffic.{json,md}, _pmc_mfma.md
set -x
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/${TAG}_bench_n1.log 2>&1; tail -1 gpurun_out/${TAG}_bench_n1.log > gpurun_out/${TAG}_bench_n1.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary > /tmp/kt2.log 2>&1
grep "^{" /tmp/kt2.log | tail -1 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_under_rocprof.json
set +x
(echo "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-secondary   (the default bench workload: 10 timed + 3 warm-up iterations + the two untimed roofline iterations; rows per kernel and launch grid)"; echo; python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt2 -name "*.db" | head -1) --md --by-grid) > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench_kernel_stats.md
set -x
cd $GRAFT_REPO_ROOT && bash tools/pmc_traffic.sh > /dev/null 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json; cp gpurun_out/pmc_traffic.md gpurun_out/${TAG}_pmc_traffic.md
bash tools/pmc_mfma.sh > /dev/null 2>&1
cp gpurun_out/pmc_mfma.md gpurun_out/${TAG}_pmc_mfma.md 2>/dev/null
ls -la gpurun_out/
cut -c1-400 gpurun_out/${TAG}_bench_n1.json
