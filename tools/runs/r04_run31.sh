cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_bench_shapes.py tests/test_gpu_gae_env_optim.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3
timeout 200 python tools/sac_bench.py 2>&1 | grep -E "SAC" | tail -1
timeout 200 python tools/sac_bench.py full_jit 2>&1 | grep -E "SAC" | tail -1
