"""One context repeats the minibatch gradient while a second stream runs unrelated torch work (or a second context)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from rlx_amd.hip import Ctx, PpoHparams
import test_gpu_dist as T
dev = torch.device("cuda:0")
Tn, NG = 128, 4096
ps, cs, pd, cd, P0, C0 = T._nets(dev, seed=2)
S, Ac, LP, R, AD = T._rollout(dev, Tn, NG, seed=2)
hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)
ctx = Ctx(0)
other = torch.cuda.Stream()
kind = sys.argv[1]
X = torch.randn(8192, 8192, device=dev)
big = torch.empty(64 << 20, device=dev)
mb = 32768
g = torch.Generator(device=dev); g.manual_seed(mb)
idx = torch.randperm(Tn * NG, device=dev, generator=g)[:mb].to(torch.int32).contiguous()
ref, nbad = None, 0
for rep in range(60):
    with torch.cuda.stream(other):
        if kind == "matmul":
            for _ in range(2): Y = X @ X
        elif kind == "fill":
            for _ in range(20): big.fill_(1.0)
        elif kind == "copy":
            for _ in range(10): big2 = big.clone()
    gp, gc, m = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.empty(8, device=dev)
    ctx.ppo_minibatch_fwd_bwd(pd, P0, gp, cd, C0, gc, m, S, Ac, LP, R, AD, idx, hp, mb_global=32768, phase=2)
    torch.cuda.synchronize()
    if ref is None:
        ref = (gp.clone(), gc.clone())
    elif not (torch.equal(gp, ref[0]) and torch.equal(gc, ref[1])):
        nbad += 1
print(f"next to '{kind}': {nbad} differing results of 59")
