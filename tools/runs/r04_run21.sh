cd $GRAFT_REPO_ROOT
for v in roabl1 roabl2 roabl3; do echo -n "$v: "; RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/librlxhip_$v.so timeout 120 python tools/section_times.py 2>&1 | grep "iter 2" | cut -c1-60; done
