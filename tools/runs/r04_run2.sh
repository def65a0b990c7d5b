# round 4, GPU call 2: fp16x3 engine with activation scale 16 -- whole GPU suite (all failures), default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_multiprocess.py > gpurun_out/r2/pytest_gpu.log 2>&1
grep -E "passed|failed|error" gpurun_out/r2/pytest_gpu.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r2/pytest_gpu.log | head -40
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2/bench.log 2>&1; tail -1 gpurun_out/r2/bench.log > gpurun_out/r2/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')})
for r in d['roofline']['per_shape']: print(r['kernel'], r['shape_MNK'], r['avg_launch_us'], r['frac'])
print({k:(v.get('value'), v.get('ms_per_update', v.get('ms_per_step'))) for k,v in d.get('secondary_configs',{}).items()})
iso=d['roofline'].get('isolated') or {}
print({k:v.get('avg_launch_us') for k,v in (iso.get('per_kernel') or {}).items()})
PY
