cd $GRAFT_REPO_ROOT
timeout 200 python tools/sac_bench.py full_jit 2>&1 | grep "updates/s" | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktj; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktj -- python $GRAFT_REPO_ROOT/tools/sac_bench.py full_jit > /tmp/ktj.log 2>&1 < /dev/null
DB=$(find /tmp/ktj -name "*.db" | head -1)
if [ -n "$DB" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB --md | head -30 | cut -c1-130; python $GRAFT_REPO_ROOT/tools/rocpd_timeline.py $DB 0.6 140 > $GRAFT_REPO_ROOT/gpurun_out/r04_sac_fulljit_timeline.txt 2>&1; fi
