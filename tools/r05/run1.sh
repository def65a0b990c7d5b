# round 5, GPU call 1: new tests first, then the regimes the twin schedule touches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
timeout 900 python -m pytest tests/test_gpu_twin_update.py tests/test_gpu_bench_shapes.py tests/test_gpu_errors.py tests/test_gpu_full_size.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r1_tests_new.log 2>&1
tail -25 gpurun_out/r1_tests_new.log
for s in "ppo_twin=0" "ppo_twin=1"; do timeout 300 python tools/update_host_time.py --mb 4096 $s; done > gpurun_out/r1_host_time.log 2>&1
timeout 300 python tools/update_host_time.py --mb 32768 "ppo_twin=0" "ppo_twin=1" >> gpurun_out/r1_host_time.log 2>&1
cat gpurun_out/r1_host_time.log
timeout 300 python bench.py --minibatch-size-global 4096 --no-prof --steps 5 --warmup 2 > gpurun_out/r1_mb4096_twin.log 2>&1; tail -1 gpurun_out/r1_mb4096_twin.log
timeout 300 python bench.py --minibatch-size-global 4096 --no-prof --steps 5 --warmup 2 --lib-option ppo_twin=0 > gpurun_out/r1_mb4096_chain.log 2>&1; tail -1 gpurun_out/r1_mb4096_chain.log
timeout 300 python bench.py --no-prof --steps 10 --warmup 3 > gpurun_out/r1_default.log 2>&1; tail -1 gpurun_out/r1_default.log
timeout 300 python bench.py --no-prof --steps 10 --warmup 3 --lib-option ppo_twin=1 > gpurun_out/r1_default_twin.log 2>&1; tail -1 gpurun_out/r1_default_twin.log
timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_twin_update.py --deselect tests/test_gpu_bench_shapes.py --deselect tests/test_gpu_errors.py --deselect tests/test_gpu_full_size.py --deselect tests/test_gpu_dist.py > gpurun_out/r1_tests_rest.log 2>&1
tail -15 gpurun_out/r1_tests_rest.log
