cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fwd2h.py -q -m gpu -x > gpurun_out/r23_tests.log 2>&1
grep -v amdgpu.ids gpurun_out/r23_tests.log | tail -5
timeout 300 python tools/sac_host_time.py > gpurun_out/r23_sac.log 2>&1
grep -v amdgpu.ids gpurun_out/r23_sac.log
bash tools/sac_timeline.sh | grep -v amdgpu.ids | tail -45 | cut -c1-70
