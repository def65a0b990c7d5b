# round 4, GPU call 6: recurrent update with the out-of-place torso backward (weight gradients under the BPTT kernel)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_ppo_lstm.py tests/test_gpu_full_size_recurrent.py tests/test_gpu_gemm.py -q -x > gpurun_out/r6/pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/r6/pytest.log | tail -3; grep -E "^E  " gpurun_out/r6/pytest.log | head -20
timeout 300 python tools/ppo_lstm_bench.py 2>&1 | tail -8
RLX_OPTS=two_streams=0 timeout 300 python tools/ppo_lstm_bench.py 2>&1 | tail -3
