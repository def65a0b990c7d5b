"""Run the same minibatch gradient repeatedly (two contexts on two streams, concurrently) and report any run-to-run difference."""
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
conc = len(sys.argv) > 1 and sys.argv[1] == "conc"
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    for c in ctxs:
        c.set_option(k, int(v))
for mb in (32768,):
    g = torch.Generator(device=dev); g.manual_seed(mb)
    idx = torch.randperm(Tn * NG, device=dev, generator=g)[:mb].to(torch.int32).contiguous()
    ref = None
    nbad = 0
    for rep in range(40):
        outs = []
        for c in range(2 if conc else 1):
            with torch.cuda.stream(streams[c]):
                gp, gc, m = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.empty(8, device=dev)
                ctxs[c].ppo_minibatch_fwd_bwd(pd, P0, gp, cd, C0, gc, m, S, Ac, LP, R, AD, idx, hp, mb_global=32768, phase=2)
                outs.append((gp, gc))
        torch.cuda.synchronize()
        for gp, gc in outs:
            if ref is None:
                ref = (gp.clone(), gc.clone())
            elif not (torch.equal(gp, ref[0]) and torch.equal(gc, ref[1])):
                nbad += 1
                if nbad <= 3:
                    dp = (gp - ref[0]).abs(); dc = (gc - ref[1]).abs()
                    def segs(spec, d):
                        out, off, inn = [], 0, T.O
                        for l, h in enumerate(spec.hidden):
                            names = [(f"W{l}", inn * h), (f"b{l}", h)] + ([("g", h), ("be", h)] if l == 0 else [])
                            for nm, n in names:
                                out.append(f"{nm}:{int((d[off:off + n] > 0).sum())}"); off += n
                            inn = h
                        out.append(f"rest:{int((d[off:] > 0).sum())}")
                        return " ".join(out)
                    print("    policy", segs(ps, dp), "| critic", segs(cs, dc))
                    print(f"  mb={mb} rep={rep}: policy diff max {dp.max().item():.3e} at {dp.argmax().item()} nnz {(dp > 0).sum().item()}; "
                          f"critic diff max {dc.max().item():.3e} at {dc.argmax().item()} nnz {(dc > 0).sum().item()}")
    print(f"mb={mb} concurrent={conc}: {nbad} differing results of {40 * (2 if conc else 1) - 1}")
