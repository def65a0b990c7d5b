# round 4, GPU call 1: the fp16x3 split engine -- whole GPU suite, isolated GEMM times, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_multiprocess.py > gpurun_out/r1/pytest_gpu.log 2>&1
grep -E "passed|failed|error" gpurun_out/r1/pytest_gpu.log | tail -3
timeout 300 python tools/gemm_bench.py > gpurun_out/r1/gemm_bench.log 2>&1; cat gpurun_out/r1/gemm_bench.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r1/bench.log 2>&1; tail -1 gpurun_out/r1/bench.log > gpurun_out/r1/bench.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r1/bench.json'))
print({k:d[k] for k in ('value','ms_per_step')})
print(json.dumps(d.get('roofline',{}).get('per_shape',[]))[:3000])
print(json.dumps(d.get('secondary_configs',{}))[:1500])
PY
