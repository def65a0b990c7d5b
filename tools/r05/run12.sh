cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
timeout 300 python tools/update_host_time.py --mb 32768 "lf_idle_cus=32" "lf_idle_cus=51" "lf_idle_cus=0" "lf_idle_cus=85" "lf_idle_cus=32" > gpurun_out/r12_lf.log 2>&1
grep -v amdgpu.ids gpurun_out/r12_lf.log
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r12_tests_all.log 2>&1
tail -5 gpurun_out/r12_tests_all.log
