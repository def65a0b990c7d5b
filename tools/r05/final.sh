cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/prof_all.sh r05f > gpurun_out/r05f_prof_all.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --minibatch-size-global 4096 --no-cpu-baseline --no-secondary 2>&1 | tail -1 > gpurun_out/r05f_bench_mb4096.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-prof 2>&1 | tail -1 > gpurun_out/r05f_bench_noprof.json
bash tools/ppo_timeline.sh > /dev/null 2>&1; cp gpurun_out/ppo_timeline.txt gpurun_out/r05f_ppo_timeline.txt
bash tools/smallmb_prof.sh > /dev/null 2>&1; cp gpurun_out/smallmb_timeline.txt gpurun_out/r05f_ppo_timeline_mb4096.txt; cp gpurun_out/smallmb_kernel_stats.md gpurun_out/r05f_mb4096_kernel_stats.md
bash tools/sac_timeline.sh > /dev/null 2>&1; cp gpurun_out/sac_timeline.txt gpurun_out/r05f_sac_timeline.txt
cut -c1-300 gpurun_out/r05f_bench_n1.json; echo; cut -c1-200 gpurun_out/r05f_bench_mb4096.json; echo; cut -c1-200 gpurun_out/r05f_bench_noprof.json
