cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c90-200
RLX_REPRO_PACKED_F32=1 python rl-x_amd/build.py --force 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | cut -c90-200
for m in 0 1 2 3 4 5; do python tools/probes/l1fwd_victim.py $m 2>&1 | tail -1; done
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 --deselect tests/test_isa_device_code.py 2>&1 | tail -5
timeout 300 python -m pytest tests/test_gpu_dist.py tests/test_gpu_full_size.py -m gpu -q --timeout 600 2>&1 | tail -2
timeout 300 python -m pytest tests/test_gpu_dist.py tests/test_gpu_full_size.py -m gpu -q --timeout 600 2>&1 | tail -2
