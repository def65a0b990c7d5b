# per-rank workload of configs[2] on one GPU (minibatch 4096 rows, 1280 updates / iteration): option sweep
cd "$(dirname "$0")/.."
for opt in "" "--lib-option dw_overlap=1" "--lib-option dw_overlap=0" "--no-prof --lib-option graph_update=1" "--no-prof" "--lib-option chain_phase=1" "--lib-option pipeline_updates=0"; do
  echo "== mb4096 $opt"
  timeout 300 python bench.py --steps 5 --warmup 3 --minibatch-size-global 4096 --no-cpu-baseline --no-secondary $opt 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print(d['ms_per_step'], 'ms', d['value'])
"
done
