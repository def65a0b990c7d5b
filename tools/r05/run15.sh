cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
mkdir -p gpurun_out
timeout 300 python tools/update_host_time.py --mb 32768 "ppo_twin=0" "ppo_twin=1" "ppo_twin=0" "ppo_twin=1" > gpurun_out/r15_twin.log 2>&1
timeout 300 python tools/update_host_time.py --mb 16384 "ppo_twin=0" "ppo_twin=1" >> gpurun_out/r15_twin.log 2>&1
timeout 300 python tools/update_host_time.py --mb 8192 "ppo_twin=0" "ppo_twin=1" >> gpurun_out/r15_twin.log 2>&1
grep -v amdgpu.ids gpurun_out/r15_twin.log
