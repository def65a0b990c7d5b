import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import nets
from oracle.sharding import local_minibatches
from rlx_amd.hip import Ctx, PpoHparams, mlp_desc
from rlx_amd.hip import lib as L
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_dist as T
dev = torch.device("cuda:0")
ctx = Ctx(0)
Tn, NG, mb, E, world, rank = 8, 64, 32, 2, 2, 0
ps, cs, pd, cd, P0, C0 = T._nets(dev, seed=2)
S, Ac, LP, R, AD = T._rollout(dev, Tn, NG, seed=2)
hp = PpoHparams(0.1, 0.01, 0.7, 5.0, 0.9, 0.999, 1e-8)
n_upd = E * (Tn * NG // mb)
lr = np.full(n_upd, 4e-4, np.float32)
key = L.prng_key(3)
perm = torch.empty(E * Tn * NG, dtype=torch.int32, device=dev)
ctx.permutation(key, perm, E, Tn * NG)
nl = NG // world
shards = [tuple(x[:, r * nl:(r + 1) * nl].contiguous() for x in (S, Ac, LP, R, AD)) for r in range(world)]
rows = [local_minibatches(perm.cpu(), n_upd, mb, NG, nl, r * nl) for r in range(world)]
a_all = AD.view(-1)[perm.long()].double().view(n_upd, mb)
stats_g = torch.stack([a_all.sum(1), (a_all * a_all).sum(1), torch.full((n_upd,), float(mb), device=dev, dtype=torch.float64),
                       torch.zeros(n_upd, device=dev, dtype=torch.float64)], dim=1).contiguous()
me = Ctx(0); me.set_rank(rank, world)
aux = (Ctx(0), Ctx(0))
side = me.side_stream()
P, C, met = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
st = {"p": 0, "c": 0}
def hook(ptr, n, dtype, on_side):
    if dtype == 1:
        buf = T._view(ptr, n, 1, dev).view(n_upd, 4)
        torch.cuda.synchronize()
        print("local stats[0]", buf[0].tolist(), "global", stats_g[0].tolist())
        buf.copy_(stats_g)
        torch.cuda.synchronize()
        chk = T._view(ptr, n, 1, dev).view(n_upd, 4)
        print("alias write ok:", bool(torch.equal(chk, stats_g)))
        return
    if n == n_upd * 10:
        return
    which = "c" if on_side else "p"
    u = st[which]; st[which] += 1
    buf = T._view(ptr, n, 0, dev)
    s_ = side if on_side else torch.cuda.current_stream()
    with torch.cuda.stream(s_):
        tot = torch.zeros_like(buf)
        for r in range(world):
            comp, cnt, off = rows[r]
            idx = comp[off[u]:off[u + 1]].to(dev)
            g_p, g_c, m = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.empty(8, device=dev)
            if idx.numel():
                aux[on_side].ppo_minibatch_fwd_bwd(pd, P, g_p, cd, C, g_c, m, *shards[r], idx, hp, mb_global=mb,
                                                   stats_io=stats_g[u].clone(), phase=2)
            g = g_c if on_side else g_p
            if r == rank and u < 3:
                s_.synchronize()
                err = ((buf - g).norm() / g.norm()).item()
                print(f"u={u} {which}: local |buf|={buf.norm().item():.5e} |aux|={g.norm().item():.5e} rel err {err:.3e} cnt={int(cnt[u])}")
            tot += g
        buf.copy_(tot)
me.set_allreduce_hook(hook)
z = lambda x: torch.zeros_like(x)
me.ppo_update_dist(pd, P, z(P), z(P), cd, C, z(C), z(C), *shards[rank], NG, rank * nl, E, mb, key, 0, lr, hp, met)
torch.cuda.synchronize()
Pr, Cr, metr = P0.clone(), C0.clone(), torch.empty(n_upd, 10, device=dev)
ctx.ppo_update(pd, Pr, z(Pr), z(Pr), cd, Cr, z(Cr), z(Cr), S, Ac, LP, R, AD, E, mb, key, 0, lr, hp, metr)
torch.cuda.synchronize()
print("final dP max", (P - Pr).abs().max().item(), "frac ok", ((P - Pr).abs() <= 2e-5 + 1e-3 * Pr.abs()).float().mean().item())
print("met", met[:3, :2].tolist(), metr[:3, :2].tolist())
print("norms", met[:3, 8:].tolist(), metr[:3, 8:].tolist())
