# k_rollout_step by phase: the kernel's average duration (rocprofv3 kernel trace) when it returns after layer 0 / the hidden layers /
# the head / not at all (option "ro_exit" = 1 / 2 / 3 / 0).   bash tools/rollout_phases.sh   -> gpurun_out/rollout_phases.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for e in 1 2 3 0; do
  rm -rf /tmp/rp$e
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp$e -- python $GRAFT_REPO_ROOT/tools/rollout_phases.py $e > /tmp/rp$e.log 2>&1
  echo "ro_exit=$e: $(python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/rp$e -name '*.db' | head -1) --md | grep k_rollout_step)"
done | tee $GRAFT_REPO_ROOT/gpurun_out/rollout_phases.txt
