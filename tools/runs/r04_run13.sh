cd /tmp && export TMPDIR=/tmp
for mb in 4096 32768; do
rm -rf /tmp/kt$mb; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt$mb -- python $GRAFT_REPO_ROOT/tools/update_host_time.py --mb $mb --reps 1 > /tmp/kt$mb.log 2>&1
DB=$(find /tmp/kt$mb -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $DB --md | head -16 | cut -c1-110
done
