cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 200 python tools/sac_bench.py 2>&1 | grep -E "SAC" | tail -1
timeout 200 python tools/sac_bench.py bx_debug=512 2>&1 | grep -E "SAC" | tail -1
done
timeout 200 python tools/sac_bench.py full_jit 2>&1 | grep -E "SAC" | tail -1
timeout 200 python tools/sac_bench.py full_jit bx_debug=512 2>&1 | grep -E "SAC" | tail -1
