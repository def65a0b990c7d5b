cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -m gpu -q -s -k configs3 2>&1 | grep -E "configs|blocks|nkinks|passed|failed|Error" | cut -c1-1500
