cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fwd2h.py -q -m gpu -x > gpurun_out/r24_tests.log 2>&1
grep -v amdgpu.ids gpurun_out/r24_tests.log | tail -15
python tools/fwd2h_phases.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/sac_host_time.py > gpurun_out/r24_sac.log 2>&1
grep -v amdgpu.ids gpurun_out/r24_sac.log
