set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_n1.log 2>&1; tail -1 gpurun_out/bench_n1.log > gpurun_out/bench_n1.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /tmp/kt2.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt2 -name "*.db" | head -1) --md > $GRAFT_REPO_ROOT/gpurun_out/bench_kernel_stats.md 2>&1
cd $GRAFT_REPO_ROOT && bash tools/pmc_traffic.sh > /dev/null 2>&1
bash tools/pmc_mfma.sh > /dev/null 2>&1
ls -la gpurun_out/
