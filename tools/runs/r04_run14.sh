cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | tail -45
import cProfile, pstats, sys, io
sys.argv = ["sac_bench.py"]
import runpy
pr = cProfile.Profile()
pr.enable()
runpy.run_path("tools/sac_bench.py", run_name="__main__")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats("sac.py|random_obs|lib.py|copy_|method", 30)
print(s.getvalue())
PY
