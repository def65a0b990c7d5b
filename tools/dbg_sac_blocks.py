#!/usr/bin/env python
"""Per-parameter-block gradient error of ONE SAC update against the float64 oracle, over a sweep of library options:
    python tools/dbg_sac_blocks.py [O A B H] [name=value ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import prng, sac  # noqa: E402
from rlx_amd.hip import Ctx, SacHparams, mlp_desc  # noqa: E402

nums = [int(x) for x in sys.argv[1:] if x and "=" not in x]
O, A, B, H = nums if len(nums) == 4 else (17, 6, 256, 256)
settings = [x for x in sys.argv[1:] if "=" in x or not x] or [""]
dev = torch.device("cuda:0")
ctx = Ctx(0)
_t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def blocks(spec):
    out = []
    for i, l in enumerate(spec.layers):
        out += [(f"W{i}", l["W"], l["in"] * l["out"]), (f"b{i}", l["b"], l["out"])]
    h = spec.head
    return out + [("Wh", h["W"], h["in"] * h["out"]), ("bh", h["b"], h["out"])]


for noise in (0.0, 0.02):
    rng = np.random.default_rng(5)
    ps, qs = sac.make_specs(O, A, H)
    pp = (sac.lecun_normal_init(ps, rng) + noise * rng.standard_normal(ps.n_params)).astype(np.float32)
    pp[ps.head["W"]:ps.head["W"] + ps.head["in"] * ps.head["out"]] *= 0.1
    qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + noise * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    qtp = qp.copy()
    s, s2 = rng.standard_normal((B, O)).astype(np.float32), rng.standard_normal((B, O)).astype(np.float32)
    a = (rng.random((B, A)) * 2 - 1).astype(np.float32)
    r, term = rng.standard_normal(B).astype(np.float32), (rng.random(B) < 0.1).astype(np.float32)
    key = prng.prng_key(11)
    f = lambda x: x.astype(np.float64)
    _, e1, e2 = sac.sample_noise(key, B, A, True)
    met_e, gp_e, gq_e, ga_e = sac.loss_and_grads(ps, f(pp), qs, f(qp), f(qtp), np.float64(0.0), f(s), f(s2), f(a), f(r), f(term),
                                                 f(e1), f(e2), 0.99, -float(A))
    pd = mlp_desc(ps.in_dim, ps.hidden, ps.out_dim, ps.act, ps.ln_first, False)
    qd = mlp_desc(qs.in_dim, qs.hidden, qs.out_dim, qs.act, qs.ln_first, False)
    for setting in settings:
        for kv in [x for x in setting.split(",") if x]:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
        res = []
        for rep in range(2):
            P, Q, QT, LA = _t(pp), _t(qp), _t(qtp), torch.zeros(1, device=dev)
            pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
            am, av, met = torch.zeros(1, device=dev), torch.zeros(1, device=dev), torch.zeros(10, device=dev)
            hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 3e-4, 3e-4, 0.9, 0.999, 1e-8)
            ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, (_t(s), _t(s2), _t(a), _t(r), _t(term)), key, 0, hp, met, 1)
            torch.cuda.synchronize()
            res.append((pm.cpu().numpy() * 10, qm.cpu().numpy() * 10))
        gp_d, gq_d = res[0]
        n = qs.n_params
        rel = lambda d, e: float(f"{np.linalg.norm(d - e) / max(np.linalg.norm(e), 1e-30):.1e}")
        print(f"noise {noise} [{setting or 'defaults'}] repeat-identical {all(np.array_equal(x, y) for x, y in zip(*res))} "
              f"policy {rel(gp_d, gp_e)} critic {rel(gq_d, gq_e)}")
        print("   policy:", {nm: rel(gp_d[o:o + ln], gp_e[o:o + ln]) for nm, o, ln in blocks(ps)})
        for k in range(2):
            print(f"   q{k}:", {nm: rel(gq_d[k * n + o:k * n + o + ln], gq_e[k * n + o:k * n + o + ln]) for nm, o, ln in blocks(qs)})
