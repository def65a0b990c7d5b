"""Debug aid: four SAC act + update rounds with kept / fresh weight images, twice each: which runs agree bit for bit?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import sac, prng
from rlx_amd.hip import Ctx, SacHparams
import test_gpu_sac as T
dev = torch.device("cuda:0")
ctx = Ctx(0)
O, A, B, H = 376, 17, 4096, 256
rng = np.random.default_rng(9)
ps, qs = sac.make_specs(O, A, H)
pp = (sac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
data = [rng.standard_normal((B, O)), rng.standard_normal((B, O)), np.tanh(rng.standard_normal((B, A))), rng.standard_normal(B), (rng.random(B) < 0.2)]
obs = rng.standard_normal((B, O))
pd, qd = T._descs(ps, qs)
hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 3e-4, 3e-4, 0.9, 0.999, 1e-8, 0)
res = {}
for tag, keep in (("k1a", 1), ("k1b", 1), ("k0a", 0), ("k0b", 0)):
    hp.keep_images = keep
    ctx.sac_invalidate_images()
    P, Q, QT = T._t(pp, dev), T._t(qp, dev), T._t(qp, dev)
    LA = T._t(np.array([-0.3]), dev)
    pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
    am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    met = torch.zeros(10, device=dev)
    batch = tuple(T._t(x, dev) for x in data)
    ob = T._t(obs, dev)
    key, akey, cnt = prng.prng_key(4), prng.prng_key(5), 0
    out = []
    for r in range(int(os.environ.get("ROUNDS", "4"))):
        act = torch.empty(B, A, device=dev)
        akey = ctx.sac_act(pd, P, ob, akey, act, -20.0, 2.0)
        key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, key, cnt, hp, met, 1)
        out += [act.cpu().numpy(), met.cpu().numpy().copy(), P.cpu().numpy().copy(), Q.cpu().numpy().copy(), QT.cpu().numpy().copy()]
    res[tag] = out
names = ["act", "met", "P", "Q", "QT"]
for a, b in (("k1a", "k1b"), ("k0a", "k0b"), ("k1a", "k0a")):
    diffs = [f"r{i // 5}:{names[i % 5]}" for i, (x, y) in enumerate(zip(res[a], res[b])) if not np.array_equal(x, y)]
    print(a, "vs", b, "first differences:", diffs[:6])
a, b = res["k1a"][4], res["k0a"][4]
d = np.nonzero(a != b)[0]
print("QT r0 differing elements:", d.size, "of", a.size, "first", d[:10], "last", d[-5:], "max abs diff", np.abs(a - b).max(),
      "rel", (np.abs(a - b) / (np.abs(b) + 1e-30)).max())
