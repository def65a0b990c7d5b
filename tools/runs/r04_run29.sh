cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_fastsac.py tests/test_gpu_train.py tests/test_gpu_reference_fixture.py tests/test_gpu_mlp.py tests/test_gpu_ppo_lstm.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('bench', d['ms_per_step'], 'ms', d['value'])
"
timeout 200 python tools/section_times.py 2>&1 | tail -8
