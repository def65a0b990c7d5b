cd $GRAFT_REPO_ROOT
echo "== default build"; python tools/gemm_bench.py 2>&1 | grep "dx_bx"; python tools/ab_option.py l1fwd_mfma 1 1 --blocks 2 2>&1 | tail -1
RLX_EXTRA_DEFINES="-DRLX_WS_MIN_WAVES=4" python rl-x_amd/build.py --force 2>&1 | tail -1
echo "== WS kernels capped at 128 VGPRs"; python tools/gemm_bench.py 2>&1 | grep "dx_bx"; python tools/ab_option.py l1fwd_mfma 1 1 --blocks 2 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_mlp.py -m gpu -q -x 2>&1 | tail -1
