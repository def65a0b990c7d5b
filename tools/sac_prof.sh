# SAC at BASELINE configs[3] shapes: parity tests, plain timing, rocprofv3 kernel stats and a one-step timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_gae_env_optim.py tests/test_gpu_sac.py tests/test_gpu_mlp.py -x -q 2>&1 | tail -3
timeout 200 python tools/sac_bench.py 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kts; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kts -- python $GRAFT_REPO_ROOT/tools/sac_bench.py > /tmp/kts.log 2>&1
DB=$(find /tmp/kts -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB --md > $GRAFT_REPO_ROOT/gpurun_out/sac_kernel_stats.md 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB 0.8 200 > $GRAFT_REPO_ROOT/gpurun_out/sac_timeline.txt 2>&1
head -40 $GRAFT_REPO_ROOT/gpurun_out/sac_kernel_stats.md
