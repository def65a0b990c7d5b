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


def _case(O, out_dim, policy, rng):
    ps, qs = sac.make_specs(O if policy else O - 17, 17, 256)
    spec = ps if policy else qs
    assert spec.in_dim == O and spec.out_dim == out_dim
    par = (sac.lecun_normal_init(spec, rng) + 0.02 * rng.standard_normal(spec.n_params)).astype(np.float32)
    return spec, par


@pytest.mark.parametrize("O,out_dim,policy", [(376, 34, True), (393, 1, False)])
@pytest.mark.parametrize("n", [4096, 4099, 8192 + 31])
def test_fused_forward_matches_float64_and_the_three_launches(ctx, dev, O, out_dim, policy, n):
    rng = np.random.default_rng(O + n)
    spec, par = _case(O, out_dim, policy, rng)
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

