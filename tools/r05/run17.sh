cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_twin_update.py tests/test_gpu_bench_shapes.py -q -m gpu -x > gpurun_out/r17_tests.log 2>&1
tail -5 gpurun_out/r17_tests.log
for lib in librlxhip.so librlxhip_pfc4.so; do
echo "== $lib" >> gpurun_out/r17_tail.log
RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/$lib timeout 300 python tools/update_host_time.py --mb 32768 "ppo_tail=1" "ppo_tail=2" "ppo_tail=1" "ppo_tail=2" >> gpurun_out/r17_tail.log 2>&1
RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/$lib timeout 300 python tools/update_host_time.py --mb 4096 "ppo_tail=1" "ppo_tail=2" >> gpurun_out/r17_tail.log 2>&1
RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/$lib timeout 300 python tools/update_host_time.py --mb 16384 "ppo_tail=1" "ppo_tail=2" >> gpurun_out/r17_tail.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r17_tail.log | awk '{print $1, $2, $3, $10, $11, $12, $13, $14}'
