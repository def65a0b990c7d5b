cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fastsac.py -m gpu -x -q 2>&1 | tail -5
bash tools/fastsac_prof.sh 2>&1 | head -30
