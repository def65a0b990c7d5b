"""Victim: context 0 repeats ONE phase of the minibatch pass.  Aggressor: context 1 runs a chosen phase on another stream."""
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
streams = (torch.cuda.current_stream(), torch.cuda.Stream())
mb = 32768
g = torch.Generator(device=dev); g.manual_seed(mb)
idx = torch.randperm(Tn * NG, device=dev, generator=g)[:mb].to(torch.int32).contiguous()
a = AD.view(-1)[idx.long()].double()
stats = torch.stack([a.sum(), (a * a).sum(), torch.tensor(float(mb), device=dev, dtype=torch.float64), torch.zeros((), device=dev, dtype=torch.float64)])
def run(c, phase):
    gp, gc, m = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.empty(8, device=dev)
    ctxs[c].ppo_minibatch_fwd_bwd(pd, P0, gp, cd, C0, gc, m, S, Ac, LP, R, AD, idx, hp, mb_global=mb, stats_io=stats.clone(), phase=phase)
    return gp, gc
vm, am = int(sys.argv[1]), int(sys.argv[2])
ctxs[0].set_option("bx_debug", vm)
ctxs[1].set_option("bx_debug", am)
print("victim mask", vm, "aggressor mask", am)
for c in range(2):
    run(c, 5)        # gather rows once per context
torch.cuda.synchronize()
for vic, agg in ((4, 4), (6, 6)):
    ref, bad = None, 0
    for rep in range(40):
        with torch.cuda.stream(streams[1]):
            for _ in range(2):
                run(1, agg)
        with torch.cuda.stream(streams[0]):
            outs = [run(0, vic) for _ in range(2)]
        torch.cuda.synchronize()
        for gp, gc in outs:
            o = gc if vic == 4 else gp
            if ref is None:
                ref = o.clone()
            elif not torch.equal(o, ref):
                bad += 1
    print(f"victim phase {vic} ({'critic' if vic == 4 else 'policy'}), aggressor phase {agg}: {bad} differing of 79")

regs = []
for c in range(2):
    for bank in range(2):
        for slot in range(60):
            try:
                ptr = ctxs[c].get_counter(f"scratch_ptr:{bank}:{slot}")
                nb = ctxs[c].get_counter(f"scratch_bytes:{bank}:{slot}")
            except Exception:
                break
            if ptr:
                regs.append((ptr, nb, c, bank, slot))
regs.sort()
for i in range(len(regs) - 1):
    if regs[i][0] + regs[i][1] > regs[i + 1][0]:
        print("OVERLAP", regs[i], regs[i + 1])
print(len(regs), "scratch regions checked")
