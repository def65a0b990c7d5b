# round 4, GPU call 4: k_dx_l1bwd with the z1 recompute and dW1 on the fp16 pipe -- parity tests, isolated times (PFX 4 vs 2), bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_bench_shapes.py tests/test_gpu_full_size.py tests/test_gpu_gemm.py -q -x > gpurun_out/r4/pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r4/pytest.log | tail -3; grep -E "^E  " gpurun_out/r4/pytest.log | head -20
for tag in "" _pfx2; do
  echo "=== lib$tag"
  RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/librlxhip$tag.so timeout 200 python tools/mb_bench.py --reps 20 2>&1 | grep -E "^==|k_dx_l1bwd"
  RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/librlxhip$tag.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench ms_per_step', d['ms_per_step'])"
done 2>&1 | tee gpurun_out/r4/variants.log
