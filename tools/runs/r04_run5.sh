# round 4, GPU call 5: where the other targets stand with the fp16x3 engine
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5
echo "--- small-minibatch regime (per-rank work of configs[2])"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --minibatch-size-global 4096 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mb4096 ms_per_step', d['ms_per_step'])"
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --minibatch-size-global 4096 --no-prof 2>/dev/null | tail -1
echo "--- default"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r5/bench.log 2>&1; tail -1 gpurun_out/r5/bench.log > gpurun_out/r5/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')})
print({k:v for k,v in d.items() if k in ('phases','phase_ms','sections')})
print([k for k in d.keys()])
PY
timeout 200 python tools/section_times.py 2>&1 | tail -15
