timeout 300 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -5
P="import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['isolated']['all_mfma_kernels'])"
echo "== plain (default)"
timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>&1 | tail -1 | python -c "$P"
echo "== pipelined"
timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --lib-option l1bwd_pipelined=1 2>&1 | tail -1 | python -c "$P"
echo "== arch flax plain / pipelined"
timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --arch flax 2>&1 | tail -1 | python -c "$P"
timeout 200 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --arch flax --lib-option l1bwd_pipelined=1 2>&1 | tail -1 | python -c "$P"
