cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sac.py tests/test_gpu_fwd2h.py tests/test_gpu_bench_shapes.py -q -m gpu -x > gpurun_out/r27_tests.log 2>&1
grep -v amdgpu.ids gpurun_out/r27_tests.log | tail -8
python tools/fwd2h_phases.py 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python tools/sac_host_time.py > gpurun_out/r27_sac.log 2>&1
grep -v amdgpu.ids gpurun_out/r27_sac.log
