"""cProfile of the SAC per-step host code (tools/sac_bench.py loop): where the submission time goes on the Python side."""
import cProfile, pstats, sys, os, io
sys.argv = [sys.argv[0]] + sys.argv[1:]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "tools", "sac_bench.py")).read().split("opts = dict(")[0]
exec(compile(src.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(ROOT)), "sac_bench_head", "exec"))
for _ in range(30): state = vector_step(state)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300): state = vector_step(state)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue())
