"""GPU: k_fwd2h (fwd2h.hip) -- the whole forward of a 256-256 network incl. its head in one launch per 32-row tile, the kernel the
SAC step's forward passes take at batches >= 4096 -- against the float64 oracle and against the three launches it replaces."""
import numpy as np
import pytest
import torch

from oracle import nets, sac
from rlx_amd.hip import mlp_desc

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)


def _case(O, out_dim, policy, rng, arch="flax"):
    ps, qs = sac.make_specs(O if policy else O - 17, 17, 256, arch=arch)
    spec = ps if policy else qs
    assert spec.in_dim == O and spec.out_dim == out_dim
    par = (sac.lecun_normal_init(spec, rng) + 0.02 * rng.standard_normal(spec.n_params)).astype(np.float32)
    return spec, par


@pytest.mark.parametrize("arch", ["flax", "full_jit"])
@pytest.mark.parametrize("O,out_dim,policy", [(376, 34, True), (393, 1, False)])
@pytest.mark.parametrize("n", [4096, 4099, 8192 + 31])
def test_fused_forward_matches_float64_and_the_three_launches(ctx, dev, O, out_dim, policy, n, arch):
    """flax: 256-256 ReLU nets (k_fwd2h); full_jit: 512-LayerNorm-256-128 ELU nets (k_fwd3h, replaces five launches)."""
    rng = np.random.default_rng(O + n)
    spec, par = _case(O, out_dim, policy, rng, arch)
    x = (rng.standard_normal((n, O)) * rng.choice([0.1, 1.0, 3.0], size=(1, O))).astype(np.float32)
    exp, _ = nets.forward(spec, par.astype(np.float64), x.astype(np.float64))
    d = mlp_desc(spec.in_dim, spec.hidden, spec.out_dim, spec.act, spec.ln_first, False)
    outs = {}
    for on in (1, 0):
        ctx.set_option("fwd2h", on)
        try:
            out = torch.full((n, out_dim), float("nan"), device=dev)
            ctx.prof_begin()
            ctx.mlp_fwd(d, _t(par, dev), _t(x, dev), out)
            ctx.prof_end()
            ran = {r["kernel"] for r in ctx.prof_rows()}
            assert ("k_fwd2h" in ran) == bool(on), ran
            assert ("k_gemm_fwd" in ran) == (not on), ran
            outs[on] = out.cpu().numpy()
        finally:
            ctx.set_option("fwd2h", 1)
    scale = np.abs(exp).max()
    for on in (1, 0):
        assert np.all(np.isfinite(outs[on]))
        assert np.abs(outs[on] - exp).max() <= 1e-5 * max(1.0, scale), (on, np.abs(outs[on] - exp).max(), scale)
    # the same split products in the same k order, the same fp32 head sums up to their grouping
    assert np.abs(outs[1] - outs[0]).max() <= 2e-6 * max(1.0, scale)


def test_twin_critic_forward_in_the_sac_update_takes_the_fused_kernel(ctx, dev):
    """The B = 4096 update of test_gpu_sac.py (oracle parity is checked there) launches k_fwd2h for every forward pass: the two
    policy passes and the three twin-critic passes -- and no k_gemm_fwd at all."""
    import test_gpu_sac as TS
    TS.test_sac_update_matches_oracle(ctx, dev, 376, 17, 4096, 256, 1, 0.1)
    rows = ctx.prof_rows()
    n = {k: sum(r["launches"] for r in rows if r["kernel"] == k) for k in ("k_fwd2h", "k_gemm_fwd")}
    assert n["k_fwd2h"] == 5 and n["k_gemm_fwd"] == 0, n


def test_full_jit_update_takes_the_fused_forward_and_matches_the_five_launch_passes(ctx, dev):
    """The full-jit flavour's update at B = 4096 with k_fwd3h for all five forward passes against the same update on the separate
    launches (option fwd2h = 0): the same split products per layer in the same k order -- metrics and parameters agree to rounding."""
    from oracle import prng
    from rlx_amd.hip import SacHparams
    O, A, B = 376, 17, 4096
    rng = np.random.default_rng(13)
    ps, qs = sac.make_specs(O, A, 256, arch="full_jit")
    pp = (sac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    qp = (np.concatenate([sac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    data = [rng.standard_normal((B, O)), rng.standard_normal((B, O)), np.tanh(rng.standard_normal((B, A))),
            rng.standard_normal(B), (rng.random(B) < 0.2)]
    pd = mlp_desc(ps.in_dim, ps.hidden, ps.out_dim, ps.act, ps.ln_first, False)
    qd = mlp_desc(qs.in_dim, qs.hidden, qs.out_dim, qs.act, qs.ln_first, False)
    hp = SacHparams(0.99, 0.005, -float(A), -20.0, 2.0, 3e-4, 3e-4, 3e-4, 0.9, 0.999, 1e-8, 1)
    res = []
    try:
        for on in (1, 0):
            ctx.set_option("fwd2h", on)
            P, Q, QT = _t(pp, dev), _t(qp, dev), _t(qp, dev)
            LA = _t(np.array([-0.3]), dev)
            pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
            am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
            met = torch.zeros(10, device=dev)
            batch = tuple(_t(x, dev) for x in data)
            key, cnt = prng.prng_key(4), 0
            ctx.prof_begin()
            key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, key, cnt, hp, met, 1)
            ctx.prof_end()
            nk = sum(r["launches"] for r in ctx.prof_rows() if r["kernel"] == "k_fwd2h")
            assert nk == (5 if on else 0), nk
            first = met.cpu().numpy().copy()
            for _ in range(2):
                key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, batch, key, cnt, hp, met, 1)
            res.append([first, met.cpu().numpy().copy(), P.cpu().numpy(), Q.cpu().numpy(), QT.cpu().numpy(), pm.cpu().numpy(), qm.cpu().numpy()])
    finally:
        ctx.set_option("fwd2h", 1)
    f, s_ = res
    assert np.all(np.isfinite(f[0])) and np.all(np.isfinite(f[2]))
    np.testing.assert_allclose(f[0][:8], s_[0][:8], rtol=2e-5, atol=1e-6)
    for i in (5, 6):      # first-moment vectors after three updates: gradient agreement
        assert np.linalg.norm(f[i] - s_[i]) / np.linalg.norm(s_[i]) < 1e-4, i
    for i in (2, 3, 4):
        assert np.abs(f[i] - s_[i]).max() < 5e-5, i

