# round-end check: whole GPU suite, the default bench line, and the kernel table of the same command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_n1.log 2>&1; tail -1 gpurun_out/bench_n1.log > gpurun_out/bench_n1.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /tmp/kt2.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt2 -name "*.db" | head -1) --md > $GRAFT_REPO_ROOT/gpurun_out/bench_kernel_stats.md 2>&1
cut -c1-300 $GRAFT_REPO_ROOT/gpurun_out/bench_n1.json
