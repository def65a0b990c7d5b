cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7
bash tools/lstm_timeline.sh > gpurun_out/r7/lstm_tl.log 2>&1
cp gpurun_out/lstm_timeline.txt gpurun_out/r7/ 2>/dev/null
head -170 gpurun_out/r7/lstm_timeline.txt
