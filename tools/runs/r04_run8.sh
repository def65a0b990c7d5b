cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r8
timeout 600 python -m pytest tests/test_gpu_ppo_lstm.py -q -x > gpurun_out/r8/pytest.log 2>&1; grep -E "passed|failed|error" gpurun_out/r8/pytest.log | tail -3
timeout 300 python tools/ppo_lstm_bench.py 2>&1 | grep ms/iteration
timeout 300 python tools/ppo_lstm_bench.py 2>&1 | grep ms/iteration
RLX_TL_FRAC=0.55 RLX_TL_N=130 bash tools/lstm_timeline.sh > gpurun_out/r8/lstm_tl.log 2>&1
cp gpurun_out/lstm_timeline.txt gpurun_out/r8/ 2>/dev/null
grep -E "lstm_seq|gemm_dw_bx" gpurun_out/r8/lstm_timeline.txt | head -30
