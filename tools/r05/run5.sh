# round 5, GPU call 5: round artifacts
cd $GRAFT_REPO_ROOT
bash tools/prof_all.sh r05 > gpurun_out/r5_prof_all.log 2>&1
tail -3 gpurun_out/r5_prof_all.log | cut -c1-600
bash tools/ppo_timeline.sh > /dev/null 2>&1; cp gpurun_out/ppo_timeline.txt gpurun_out/r05_ppo_timeline.txt
timeout 300 python bench.py --minibatch-size-global 4096 --no-secondary --no-cpu-baseline > gpurun_out/r05_bench_mb4096.log 2>&1; tail -1 gpurun_out/r05_bench_mb4096.log > gpurun_out/r05_bench_mb4096.json; cut -c1-300 gpurun_out/r05_bench_mb4096.json
timeout 300 python bench.py --minibatch-size-global 4096 --no-prof > gpurun_out/r05_bench_mb4096_noprof.log 2>&1; tail -1 gpurun_out/r05_bench_mb4096_noprof.log
