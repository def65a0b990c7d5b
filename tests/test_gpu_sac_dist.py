"""Data-parallel SAC update (SAC bullet): each rank holds B / G rows of the global batch
(rlx_sac_hparams.batch_global / batch_row_offset), one all-reduce over [policy grads | critic grads | loss sums], the three Adam
steps and the Polyak update applied redundantly.  The reference has no multi-device SAC; the contract is that the
G-rank result equals the one-device update (sac/flax/sac.py:128-215) on the concatenated batch.

Ranks are emulated in one process through the library's all-reduce hook: rank 1 runs first and its buffer is captured, rank 0
runs with the hook adding it -- rank 0 then holds exactly what a 2-rank job leaves on every rank.  (2 processes over RCCL / gloo:
tests/test_gpu_multiprocess.py.)"""
import numpy as np
import pytest
import torch

from oracle import prng, sac
from rlx_amd.hip import Ctx, SacHparams, mlp_desc

pytestmark = pytest.mark.gpu


def _t(a, dev, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(dev)


class _Buf:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


@pytest.mark.parametrize("O,A,Bg,H,arch,world", [(11, 3, 256, 64, "flax", 2), (40, 6, 384, 64, "full_jit", 3),
                                                  (376, 17, 8192, 256, "flax", 2)])
def test_emulated_ranks_equal_the_one_device_update(dev, O, A, Bg, H, arch, world):
    rng = np.random.default_rng(O + Bg)
    ps, qs = sac.make_specs(O, A, H, arch=arch)
    pp = (sac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    pp[ps.head["W"]:ps.head["W"] + ps.head["in"] * ps.head["out"]] *= 0.1
    qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    qtp = (qp + 0.01 * rng.standard_normal(qp.shape)).astype(np.float32)
    s, s2 = rng.standard_normal((Bg, O)).astype(np.float32), rng.standard_normal((Bg, O)).astype(np.float32)
    a = np.tanh(rng.standard_normal((Bg, A))).astype(np.float32)
    r, term = rng.standard_normal(Bg).astype(np.float32), (rng.random(Bg) < 0.2).astype(np.float32)
    key = prng.prng_key(5)
    sched = 1 if arch == "full_jit" else 0
    pd = mlp_desc(ps.in_dim, ps.hidden, ps.out_dim, ps.act, ps.ln_first, False)
    qd = mlp_desc(qs.in_dim, qs.hidden, qs.out_dim, qs.act, qs.ln_first, False)

    def run(ctx, rows, hook, bg, roff):
        P, Q, QT, LA = _t(pp, dev), _t(qp, dev), _t(qtp, dev), _t(np.array([-0.3]), dev)
        pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
        am, av, met = torch.zeros(1, device=dev), torch.zeros(1, device=dev), torch.zeros(10, device=dev)
        hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 3e-4, 3e-4, 0.9, 0.999, 1e-8, sched)
        hp.batch_global, hp.batch_row_offset = bg, roff
        ctx.set_allreduce_hook(hook)
        try:
            new_key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av,
                                          tuple(_t(x[rows], dev) for x in (s, s2, a, r, term)), key, 0, hp, met, 1)
            torch.cuda.synchronize()
        finally:
            ctx.set_allreduce_hook(None)
        return dict(P=P, Q=Q, QT=QT, LA=LA, pm=pm, qm=qm, am=am, met=met, key=new_key)

    whole = run(Ctx(0), slice(0, Bg), None, 0, 0)
    Bl = Bg // world
    peers = []
    for rk in range(world - 1, 0, -1):                     # the peers first: their contribution is captured, their result dropped
        got = {}

        def capture(ptr, n, dtype, on_side, got=got):
            got["buf"] = torch.as_tensor(_Buf(ptr, n, "<f4"), device=dev).clone()
        c = Ctx(0)
        c.set_rank(rk, world)
        run(c, slice(rk * Bl, (rk + 1) * Bl), capture, Bg, rk * Bl)
        peers.append(got["buf"])
    calls = []

    def add_peers(ptr, n, dtype, on_side):
        buf = torch.as_tensor(_Buf(ptr, n, "<f4"), device=dev)
        calls.append(n)
        for pb in peers:
            buf += pb
    c0 = Ctx(0)
    c0.set_rank(0, world)
    mine = run(c0, slice(0, Bl), add_peers, Bg, 0)
    assert len(calls) == 1                                   # ONE collective per update
    assert np.array_equal(mine["key"], whole["key"])
    rel = lambda x, y: ((x - y).norm() / y.norm().clamp_min(1e-30)).item()
    errs = {k: rel(mine[k], whole[k]) for k in ("pm", "qm", "am", "P", "Q", "QT", "LA")}
    merr = ((mine["met"] - whole["met"]).abs() / whole["met"].abs().clamp_min(1e-3)).max().item()
    print(f"sharded SAC update O={O} A={A} Bg={Bg} world={world} {arch}: relative difference to the one-device update", {k: float(f"{v:.1e}") for k, v in errs.items()},
          f"metrics {merr:.1e}")
    assert max(errs["pm"], errs["qm"], errs["am"]) < 2e-6     # same per-sample terms, summed in a different fp32 order
    assert max(errs["P"], errs["Q"], errs["QT"], errs["LA"]) < 1e-6 and merr < 2e-6


def test_sharded_update_without_a_communicator_is_refused(dev):
    ps, qs = sac.make_specs(5, 2, 64)
    pd = mlp_desc(ps.in_dim, ps.hidden, ps.out_dim, ps.act, ps.ln_first, False)
    qd = mlp_desc(qs.in_dim, qs.hidden, qs.out_dim, qs.act, qs.ln_first, False)
    P, Q = torch.zeros(ps.n_params, device=dev), torch.zeros(2 * qs.n_params, device=dev)
    z = lambda *sh: torch.zeros(*sh, device=dev)
    hp = SacHparams(0.99, 0.005, -2.0, -20.0, 2.0, 3e-4, 3e-4, 3e-4, 0.9, 0.999, 1e-8, 0)
    hp.batch_global, hp.batch_row_offset = 64, 0
    with pytest.raises(RuntimeError, match="batch_global"):
        Ctx(0).sac_update(pd, P, z(ps.n_params), z(ps.n_params), qd, Q, z(2 * qs.n_params), z(2 * qs.n_params), Q.clone(), z(1), z(1), z(1),
                          (z(32, 5), z(32, 5), z(32, 2), z(32), z(32)), prng.prng_key(1), 0, hp, z(10), 1)
