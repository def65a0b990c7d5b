"""Per-block gradient errors of rlx_fastsac_critic_update_f32 / policy_update against oracle/fastsac.py at the fixture's case 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import fastsac as ofs
from rlx_amd.hip import Ctx, lnmlp_desc
from rlx_amd.hip import lib as L
import test_gpu_fastsac as T
dev = torch.device("cuda:0")
ctx = Ctx(0)
z, g, h, O, A, NA, B, pflat, qflat, clipped = T._fixture_case(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
pd, qd = lnmlp_desc(O, ofs.POLICY_HIDDEN, 2 * A), lnmlp_desc(O + A, ofs.CRITIC_HIDDEN, NA)
hp = T._hp(h, NA, clipped)
t = lambda a: T._t(a, dev)
P, Q, QT = t(pflat), t(np.concatenate(qflat[:2])), t(np.concatenate(qflat[2:]))
qm, qv, pm, pv = (torch.zeros_like(x) for x in (Q, Q, P, P))
la, am, av = t([np.float32(h["log_alpha"])]), torch.zeros(1, device=dev), torch.zeros(1, device=dev)
batch = tuple(t(g(n)) for n in ("states", "next_states", "actions", "rewards", "dones", "truncations", "n_steps"))
met = torch.zeros(8, device=dev)
e_next, e_cur = t(g("noise_next")), t(g("noise_cur"))       # (kept alive: the library holds raw pointers)
ctx.dbg_set_sac_noise(e_next, e_cur)
key, cnt = ctx.fastsac_critic_update(pd, P, qd, Q, qm, qv, QT, la, am, av, batch, t(g("action_scale")), L.prng_key(5), 0, hp, met)
b64 = tuple(np.asarray(g(n), dtype=np.float64) for n in ("states", "next_states", "actions", "rewards", "dones", "truncations", "n_steps"))
r = ofs.critic_step(pflat.astype(np.float64), *(q.astype(np.float64) for q in qflat), float(np.float32(h["log_alpha"])), O, A, NA, b64,
                    g("noise_next"), g("action_scale"), h, clipped)
gd = qm.cpu().numpy().astype(np.float64) / (1 - h["adam_beta1"])
n = qflat[0].size
for k in range(2):
    ge = r["g_q%d" % (k + 1)]
    for name, off, ln in ofs.blocks(O + A, ofs.CRITIC_HIDDEN, NA):
        d, e = gd[k * n + off:k * n + off + ln], ge[off:off + ln]
        print(f"q{k}.{name:4s} n={ln:7d} rel {np.linalg.norm(d - e) / max(np.linalg.norm(e), 1e-30):.2e}  |e| {np.linalg.norm(e):.3e} |d| {np.linalg.norm(d):.3e}")
