cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_full_size.py tests/test_gpu_reference_fixture.py tests/test_gpu_discrete.py tests/test_gpu_obs_indices.py tests/test_gpu_sharded_update.py -m gpu -q -x 2>&1 | tail -3
timeout 200 python bench.py --steps 3 --warmup 1 --minibatch-size-global 4096 --no-cpu-baseline --no-secondary --no-prof 2>&1 | tail -1 | cut -c1-200
