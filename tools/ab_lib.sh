# Same-box A/B of two builds of the library (box-to-box variance is 1-2 %, run-to-run on one box ~0.1 %): alternates the default
# bench workload between rl-x_amd/lib/librlxhip_A.so and rl-x_amd/lib/librlxhip.so (RLX_HIP_LIBRARY selects the file).
#   cp rl-x_amd/lib/librlxhip.so rl-x_amd/lib/librlxhip_A.so   (the baseline build), change the code, python rl-x_amd/build.py, then
#   gpurun -- 'bash tools/ab_lib.sh [rounds] [extra bench.py flags]'
cd $GRAFT_REPO_ROOT
R=${1:-3}; shift
for i in $(seq $R); do
  for L in librlxhip_A.so librlxhip.so; do
    ms=$(RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/$L timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-prof "$@" 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$L $ms"
  done
done
