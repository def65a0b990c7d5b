cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r11
timeout 600 python -m pytest tests/test_gpu_gemm_px.py -q > gpurun_out/r11/pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r11/pytest.log | tail -3; grep -E "^E  |^FAILED" gpurun_out/r11/pytest.log | head -30
timeout 300 python tools/gemm_bench.py 2>&1 | grep -E "_bx|_px"
