cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
mkdir -p gpurun_out
for v in "" _pk_none _pk_n15b _pk_n3b _pk_n0b _pk_n15m _pk_n15e _pk_wf _pk_vn; do
for i in 1 2 3; do
echo -n "librlxhip$v run $i: " >> gpurun_out/r19_pk.log
RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/librlxhip$v.so timeout 300 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -s -k configs3 2>&1 | grep -E "dg\|\|/\|\|g\|\| policy|passed|failed" | sed 's/(numpy.*//' | tr '\n' ' ' >> gpurun_out/r19_pk.log
echo >> gpurun_out/r19_pk.log
done
done
cat gpurun_out/r19_pk.log
