cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r12
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_obs_indices.py -q > gpurun_out/r12/pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r12/pytest.log | tail -3; grep -E "^E  |^FAILED" gpurun_out/r12/pytest.log | head -30
