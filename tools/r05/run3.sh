# round 5, GPU call 3: prefetch depth A/B, cheap dW1 scale, fixed tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
L=$GRAFT_REPO_ROOT/rl-x_amd/lib
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_errors.py tests/test_gpu_dist.py tests/test_gpu_twin_update.py tests/test_gpu_gemm.py tests/test_gpu_mlp.py -q -m gpu > gpurun_out/r3_tests.log 2>&1
tail -30 gpurun_out/r3_tests.log
( for v in "" pf1 dw0; do
  lib=$L/librlxhip${v:+_$v}.so
  echo "== $lib"
  RLX_HIP_LIBRARY=$lib timeout 300 python tools/update_host_time.py --mb 32768 "ppo_tail=1" "ppo_tail=0"
  RLX_HIP_LIBRARY=$lib timeout 300 python tools/update_host_time.py --mb 4096 "ppo_tail=1"
done ) > gpurun_out/r3_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r3_ab.log
timeout 300 python tools/gemm_bench.py > gpurun_out/r3_gemm_bench.log 2>&1; grep -v amdgpu.ids gpurun_out/r3_gemm_bench.log | tail -30
RLX_HIP_LIBRARY=$L/librlxhip_pf1.so timeout 300 python tools/gemm_bench.py > gpurun_out/r3_gemm_bench_pf1.log 2>&1; grep -v amdgpu.ids gpurun_out/r3_gemm_bench_pf1.log | tail -30
timeout 300 python bench.py --no-prof --steps 10 --warmup 3 > gpurun_out/r3_default.log 2>&1; tail -1 gpurun_out/r3_default.log
RLX_HIP_LIBRARY=$L/librlxhip_pf1.so timeout 300 python bench.py --no-prof --steps 10 --warmup 3 > gpurun_out/r3_default_pf1.log 2>&1; tail -1 gpurun_out/r3_default_pf1.log
