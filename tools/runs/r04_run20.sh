cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ht; timeout 300 rocprofv3 --hip-trace --stats -d /tmp/ht -- python $GRAFT_REPO_ROOT/tools/update_host_time.py --mb 4096 --reps 1 > /tmp/ht.log 2>&1
tail -2 /tmp/ht.log
f=$(find /tmp/ht -name "*hip_api_stats*" | head -1); echo $f; head -20 $f
ls /tmp/ht/* | head
