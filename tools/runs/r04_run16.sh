cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/r04_pytest_gpu.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/r04_pytest_gpu.log | head -20
bash tools/prof_all.sh r04 > gpurun_out/r04_prof_all.log 2>&1
cut -c1-600 gpurun_out/r04_bench_n1.json
