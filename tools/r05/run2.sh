# round 5, GPU call 2: tail kernel + dW1 operand scale: tests, then A/B timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
timeout 900 python -m pytest tests/test_gpu_twin_update.py tests/test_gpu_bench_shapes.py tests/test_gpu_errors.py tests/test_gpu_fastsac.py -q -m gpu > gpurun_out/r2_tests_new.log 2>&1
tail -40 gpurun_out/r2_tests_new.log
for s in "ppo_tail=0" "ppo_tail=1" "ppo_tail=1,ppo_twin=0" "ppo_tail=0,ppo_twin=0"; do timeout 300 python tools/update_host_time.py --mb 4096 $s; done > gpurun_out/r2_host_time.log 2>&1
timeout 300 python tools/update_host_time.py --mb 32768 "ppo_tail=0" "ppo_tail=1" >> gpurun_out/r2_host_time.log 2>&1
RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/librlxhip_dw0.so timeout 300 python tools/update_host_time.py --mb 32768 "ppo_tail=0" "ppo_tail=1" >> gpurun_out/r2_host_time.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_host_time.log
timeout 300 python bench.py --no-prof --steps 10 --warmup 3 > gpurun_out/r2_default.log 2>&1; tail -1 gpurun_out/r2_default.log
timeout 300 python bench.py --no-prof --steps 10 --warmup 3 --lib-option ppo_tail=0 > gpurun_out/r2_default_notail.log 2>&1; tail -1 gpurun_out/r2_default_notail.log
timeout 300 python bench.py --minibatch-size-global 4096 --no-prof --steps 5 --warmup 2 > gpurun_out/r2_mb4096.log 2>&1; tail -1 gpurun_out/r2_mb4096.log
timeout 900 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_dist.py tests/test_gpu_reference_fixture.py tests/test_gpu_train.py tests/test_gpu_ppo_lstm.py tests/test_gpu_full_size_recurrent.py -x -q -m gpu > gpurun_out/r2_tests_more.log 2>&1
tail -8 gpurun_out/r2_tests_more.log
