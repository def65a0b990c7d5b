# round 4, GPU call 3: k_dx_l1bwd phase ablation + prefetch depth (isolated kernel times), and the two-chain timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for tag in "" _pfx3 _pfx4 _abl1 _abl2 _abl4 _abl8 _abl12; do
  echo "=== lib$tag"
  RLX_HIP_LIBRARY=$GRAFT_REPO_ROOT/rl-x_amd/lib/librlxhip$tag.so timeout 200 python tools/mb_bench.py --reps 20 2>&1 | grep -E "^==|k_dx_l1bwd|k_l1fwd|k_head|k_reduce"
done > gpurun_out/r3/l1bwd_variants.log 2>&1
cat gpurun_out/r3/l1bwd_variants.log
bash tools/ppo_timeline.sh > /dev/null 2>&1; cp gpurun_out/ppo_timeline.txt gpurun_out/r3/; head -100 gpurun_out/r3/ppo_timeline.txt
