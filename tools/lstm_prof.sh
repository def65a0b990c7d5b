# PPO+LSTM at BASELINE configs[4] shapes: plain timing with the phase split, then rocprofv3 kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/ppo_lstm_bench.py 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktl; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktl -- python $GRAFT_REPO_ROOT/tools/ppo_lstm_bench.py > /tmp/ktl.log 2>&1
DB=$(find /tmp/ktl -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB --md > $GRAFT_REPO_ROOT/gpurun_out/lstm_kernel_stats.md 2>&1
head -45 $GRAFT_REPO_ROOT/gpurun_out/lstm_kernel_stats.md
