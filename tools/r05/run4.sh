# round 5, GPU call 4: l1fused idle-CU experiment, whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
timeout 300 python tools/update_host_time.py --mb 32768 "lf_idle_cus=0" "lf_idle_cus=16" "lf_idle_cus=32" "lf_idle_cus=64" "lf_idle_cus=0" > gpurun_out/r4_lfidle.log 2>&1
grep -v amdgpu.ids gpurun_out/r4_lfidle.log
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r4_tests_all.log 2>&1
tail -12 gpurun_out/r4_tests_all.log
