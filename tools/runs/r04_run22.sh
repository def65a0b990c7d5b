cd $GRAFT_REPO_ROOT
timeout 400 bash tools/ppo_timeline.sh > /dev/null 2>&1 < /dev/null; cp gpurun_out/ppo_timeline.txt gpurun_out/r04_ppo_timeline.txt
timeout 400 bash tools/sac_timeline.sh > /dev/null 2>&1 < /dev/null; cp gpurun_out/sac_timeline.txt gpurun_out/r04_sac_timeline.txt
RLX_TL_FRAC=0.62 timeout 400 bash tools/lstm_timeline.sh > /dev/null 2>&1 < /dev/null; cp gpurun_out/lstm_timeline.txt gpurun_out/r04_lstm_timeline.txt
wc -l gpurun_out/r04_*timeline.txt
