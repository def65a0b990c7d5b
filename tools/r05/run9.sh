cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONPATH=$GRAFT_REPO_ROOT:$GRAFT_REPO_ROOT/rl-x_amd:$GRAFT_REPO_ROOT/tests
timeout 900 python -m pytest tests/test_gpu_twin_update.py tests/test_gpu_bench_shapes.py -q -m gpu -x > gpurun_out/r9_tests.log 2>&1
tail -5 gpurun_out/r9_tests.log
timeout 300 python tools/update_host_time.py --mb 32768 "dw_recompute=0" "dw_recompute=1" "dw_recompute=0" "dw_recompute=1" > gpurun_out/r9_rec.log 2>&1
timeout 300 python tools/update_host_time.py --mb 4096 "dw_recompute=0" "dw_recompute=1" >> gpurun_out/r9_rec.log 2>&1
grep -v amdgpu.ids gpurun_out/r9_rec.log
