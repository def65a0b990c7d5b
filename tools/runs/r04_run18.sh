cd $GRAFT_REPO_ROOT
for v in "$@"; do
for i in 1 2; do
echo -n "$v run $i: "
RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/librlxhip_$v.so timeout 300 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -s -k configs3 2>&1 | grep -E "dg\|\|/\|\|g\|\| policy" | sed 's/(numpy.*//' | head -1
done
done
