cd $GRAFT_REPO_ROOT
timeout 200 python tools/sac_bench.py full_jit 2>&1 | grep "updates/s" | tail -1 | cut -c1-200
timeout 200 python tools/sac_bench.py 2>&1 | grep "updates/s" | tail -1 | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_sac_dist.py tests/test_gpu_bench_shapes.py tests/test_gpu_obs_indices.py tests/test_obs_norm.py -m gpu -q 2>&1 | tail -3
