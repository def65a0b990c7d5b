"""Gradient of one minibatch with the split-fp16 GEMM engine vs the exact-fp32 engine, per parameter segment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from rlx_amd.hip import Ctx, PpoHparams
import test_gpu_dist as T
dev = torch.device("cuda:0")
ctx = Ctx(0)
Tn, NG = 128, 4096
ps, cs, pd, cd, P0, C0 = T._nets(dev, seed=2)
S, Ac, LP, R, AD = T._rollout(dev, Tn, NG, seed=2)
hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)
def segs(spec):
    out, off, inn = [], 0, T.O
    for l, h in enumerate(spec.hidden):
        out.append((f"W{l}", off, off + inn * h)); off += inn * h
        out.append((f"b{l}", off, off + h)); off += h
        if l == 0:
            out.append(("g", off, off + h)); off += h
            out.append(("be", off, off + h)); off += h
        inn = h
    out.append(("rest", off, spec.n_params))
    return out
import numpy as _np
from rlx_amd.hip import lib as L
if len(sys.argv) > 1:      # evolve the parameters first: argv[1] epochs of the real update (16 minibatches each)
    E = int(sys.argv[1])
    n_upd = E * (Tn * NG // 32768)
    z = lambda x: torch.zeros_like(x)
    met = torch.empty(n_upd, 10, device=dev)
    ctx.ppo_update(pd, P0, z(P0), z(P0), cd, C0, z(C0), z(C0), S, Ac, LP, R, AD, E, 32768, L.prng_key(8), 0,
                   _np.full(n_upd, 4e-4, _np.float32), hp, met)
    torch.cuda.synchronize()
    print("evolved", n_upd, "updates; last metrics", met[-1].tolist())
for mb in (32768, 4134, 4096, 4608, 5000):
    g = torch.Generator(device=dev); g.manual_seed(mb)
    idx = torch.randperm(Tn * NG, device=dev, generator=g)[:mb].to(torch.int32).contiguous()
    res = {}
    for bx in (0, 1):
        ctx.set_option("gemm_bx", bx)
        gp, gc, m = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.empty(8, device=dev)
        ctx.ppo_minibatch_fwd_bwd(pd, P0, gp, cd, C0, gc, m, S, Ac, LP, R, AD, idx, hp, mb_global=mb, phase=2)
        torch.cuda.synchronize()
        res[bx] = (gp.double(), gc.double(), m.clone())
    print(f"mb={mb}: metrics bx {res[1][2][:5].tolist()}  fp32 {res[0][2][:5].tolist()}")
    for name, spec, k in (("policy", ps, 0), ("critic", cs, 1)):
        a, b = res[1][k], res[0][k]
        line = f"  {name}: total {((a - b).norm() / b.norm()).item():.2e} |"
        for sname, lo, hi in segs(spec):
            line += f" {sname} {((a[lo:hi] - b[lo:hi]).norm() / (b[lo:hi].norm() + 1e-30)).item():.1e}"
        print(line)
