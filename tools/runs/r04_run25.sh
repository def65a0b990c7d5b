cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktf; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktf -- python $GRAFT_REPO_ROOT/tools/fastsac_bench.py > /tmp/ktf.log 2>&1 < /dev/null
DB=$(find /tmp/ktf -name "*.db" | head -1)
if [ -n "$DB" ]; then python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB --md | head -26 | cut -c1-130; fi
