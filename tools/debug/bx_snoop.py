"""Does an aggressor context's pass modify ANY scratch region of an idle victim context (or the shared inputs)?"""
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
ctxs = (Ctx(0), Ctx(0))
mb = 32768
g = torch.Generator(device=dev); g.manual_seed(mb)
idx = torch.randperm(Tn * NG, device=dev, generator=g)[:mb].to(torch.int32).contiguous()
a = AD.view(-1)[idx.long()].double()
stats = torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(mb), device=dev, dtype=torch.float64), torch.zeros((), device=dev, dtype=torch.float64)])
def run(c, phase):
    gp, gc, m = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.empty(8, device=dev)
    ctxs[c].ppo_minibatch_fwd_bwd(pd, P0, gp, cd, C0, gc, m, S, Ac, LP, R, AD, idx, hp, mb_global=mb, stats_io=stats.clone(), phase=phase)
    return gp, gc
ctxs[0].set_option("bx_debug", int(sys.argv[1])); ctxs[1].set_option("bx_debug", int(sys.argv[2]))
for c in range(2):
    run(c, 5); run(c, 6); run(c, 4)
torch.cuda.synchronize()
class _Buf:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}
def regions(c):
    out = []
    for bank in range(2):
        for slot in range(60):
            try:
                ptr = ctxs[c].get_counter(f"scratch_ptr:{bank}:{slot}"); nb = ctxs[c].get_counter(f"scratch_bytes:{bank}:{slot}")
            except Exception:
                break
            if ptr and nb >= 4:
                out.append((bank, slot, torch.as_tensor(_Buf(ptr, nb // 4), device=dev)))
    return out
vic = regions(0)
snap = [(b, s_, v.clone()) for b, s_, v in vic]
shared = [x.clone() for x in (P0, C0, S, Ac, LP, R, AD)]
for rep in range(10):
    run(1, 6); run(1, 4)
torch.cuda.synchronize()
for (b, s_, v), (_, _, old) in zip(vic, snap):
    if not torch.equal(v, old):
        d = (v != old)
        print(f"victim bank {b} slot {s_}: {int(d.sum())} of {v.numel()} words changed; first at {int(d.nonzero()[0])}")
for nm, x, old in zip("P0 C0 S Ac LP R AD".split(), (P0, C0, S, Ac, LP, R, AD), shared):
    if not torch.equal(x, old):
        print("shared input changed:", nm, int((x != old).sum()))
print("snoop done")
