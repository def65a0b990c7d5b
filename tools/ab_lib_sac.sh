# Same-box A/B of two library builds on the SAC configs[3] step (see tools/ab_lib.sh):   gpurun -- 'bash tools/ab_lib_sac.sh [rounds] [full_jit]'
cd $GRAFT_REPO_ROOT
R=${1:-3}; shift
for i in $(seq $R); do
  for L in librlxhip_A.so librlxhip.so; do
    echo "$L $(RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/$L timeout 300 python tools/sac_bench.py "$@" 2>/dev/null | grep 'updates/s' | tail -1 | cut -c1-60)"
  done
done
