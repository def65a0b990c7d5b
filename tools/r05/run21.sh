cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
mkdir -p gpurun_out
timeout 300 python tools/sac_host_time.py > gpurun_out/r21_sac.log 2>&1
timeout 300 python tools/sac_host_time.py full_jit >> gpurun_out/r21_sac.log 2>&1
grep -v amdgpu.ids gpurun_out/r21_sac.log
