# PPO+LSTM configs[4]: timeline of one sequence minibatch in the middle of the update (which kernels, how long, which queue, gaps)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl; timeout 300 rocprofv3 --kernel-trace -d /tmp/ktl -- python $GRAFT_REPO_ROOT/tools/ppo_lstm_bench.py > /tmp/ktl.log 2>&1
DB=$(find /tmp/ktl -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB ${RLX_TL_FRAC:-0.5} ${RLX_TL_N:-150} > $GRAFT_REPO_ROOT/gpurun_out/lstm_timeline.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/lstm_timeline.txt
