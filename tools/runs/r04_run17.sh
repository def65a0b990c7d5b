cd $GRAFT_REPO_ROOT
for i in 1 2; do
RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/librlxhip_${1:-vecall}.so timeout 300 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -s -k configs3 2>&1 | grep -E "passed|failed|^E  |rel|block" | head -12
done
