cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r9
bash tools/sac_timeline.sh > gpurun_out/r9/sac_tl.log 2>&1
cp gpurun_out/sac_timeline.txt gpurun_out/r9/ 2>/dev/null
head -130 gpurun_out/r9/sac_timeline.txt | cut -c1-70
