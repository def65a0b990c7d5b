cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_twin_update.py tests/test_gpu_bench_shapes.py tests/test_gpu_dist.py -q -m gpu -x > gpurun_out/r18_tests.log 2>&1
tail -5 gpurun_out/r18_tests.log
timeout 300 python tools/update_host_time.py --mb 4096 "ppo_tail=-1" "ppo_tail=1" > gpurun_out/r18_tail.log 2>&1
timeout 300 python tools/update_host_time.py --mb 8192 "ppo_tail=1" "ppo_tail=2" >> gpurun_out/r18_tail.log 2>&1
grep -v amdgpu.ids gpurun_out/r18_tail.log | awk '{print $1, $2, $3, $10, $11, $12, $13, $14}'
