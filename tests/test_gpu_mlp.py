"""GPU: MLP forward (both reference architectures) and the PPO minibatch loss/gradients
against the oracle.  fp32, tolerance 1e-5 relative (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import nets, ppo as oppo
from rlx_amd.hip import PpoHparams, mlp_desc

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _desc(spec):
    return mlp_desc(spec.in_dim, spec.hidden, spec.out_dim, spec.act, spec.ln_first, spec.has_logstd)


def _nets(arch, O, A, rng, perturb=0.05):
    ps = nets.make_spec(arch, O, A, True)
    cs = nets.make_spec(arch, O, 1, False)
    pp = nets.init_params(ps, rng, 0.01)
    cp = nets.init_params(cs, rng, 1.0)
    pp = (pp + perturb * rng.standard_normal(pp.shape)).astype(np.float32)
    cp = (cp + perturb * rng.standard_normal(cp.shape)).astype(np.float32)
    return ps, pp, cs, cp


def test_param_count_matches_oracle():
    from rlx_amd.hip import lib as L
    import ctypes
    lib = L.load_library()
    for arch, O, A in [("A", 17, 6), ("B", 17, 6), ("B", 4, 2), ("A", 33, 17)]:
        for pol in (True, False):
            s = nets.make_spec(arch, O, A if pol else 1, pol)
            d = _desc(s)
            assert lib.rlx_mlp_param_count(ctypes.byref(d)) == s.n_params
    assert nets.make_spec("B", 17, 6, True).n_params == 175244
    assert nets.make_spec("B", 17, 1, False).n_params == 174593
    assert nets.make_spec("A", 17, 6, True).n_params == 71948


@pytest.mark.parametrize("arch", ["A", "B"])
@pytest.mark.parametrize("n", [1, 64, 130, 1030, 4096, 5001])
def test_mlp_fwd_matches_oracle(ctx, dev, arch, n):
    rng = np.random.default_rng(n)
    ps, pp, cs, cp = _nets(arch, 17, 6, rng)
    x = rng.standard_normal((n, 17)).astype(np.float32)
    for spec, par in ((ps, pp), (cs, cp)):
        exp, _ = nets.forward(spec, par.astype(np.float64), x.astype(np.float64))
        out = torch.empty(n, spec.out_dim, device=dev)
        ctx.mlp_fwd(_desc(spec), _t(par, dev), _t(x, dev), out)
        np.testing.assert_allclose(out.cpu().numpy(), exp, rtol=1e-5, atol=5e-6)  # fp32 K=512 dot products + 1e-7-abs ELU


def _minibatch_case(arch, O, A, B, mb, rng):
    ps, pp, cs, cp = _nets(arch, O, A, rng)
    states = rng.standard_normal((B, O)).astype(np.float32)
    actions = rng.standard_normal((B, A)).astype(np.float32)
    mean, _ = nets.forward(ps, pp, states)
    logstd = pp[ps.logstd:ps.logstd + A][None, :]
    logp = (oppo.gaussian_log_prob(actions, mean, logstd) + 0.05 * rng.standard_normal(B)).astype(np.float32)
    returns = rng.standard_normal(B).astype(np.float32)
    adv = (rng.standard_normal(B) * 2 + 0.3).astype(np.float32)
    idx = rng.permutation(B)[:mb].astype(np.int32)
    return ps, pp, cs, cp, states, actions, logp, returns, adv, idx


@pytest.mark.parametrize("arch,O,A,B,mb", [("B", 17, 6, 1024, 256), ("A", 17, 6, 1024, 256), ("B", 17, 6, 700, 130),
                                           ("B", 4, 2, 512, 64), ("A", 31, 17, 512, 100), ("B", 17, 6, 8192, 4096)])
def test_ppo_minibatch_loss_and_grads(ctx, dev, arch, O, A, B, mb):
    rng = np.random.default_rng(B + mb)
    ps, pp, cs, cp, states, actions, logp, returns, adv, idx = _minibatch_case(arch, O, A, B, mb, rng)
    clip, ent, cc = 0.1, 0.01, 0.7
    f64 = lambda a: a.astype(np.float64)
    madv = oppo.normalize_advantages(f64(adv[idx]))
    loss_e, met_e, gp_e, gc_e = oppo.ppo_loss_and_grads(ps, f64(pp), cs, f64(cp), f64(states[idx]), f64(actions[idx]),
                                                        f64(logp[idx]), f64(returns[idx]), madv, clip, ent, cc)
    hp = PpoHparams(clip, ent, cc, 0.5, 0.9, 0.999, 1e-8)
    pg = torch.zeros(ps.n_params, device=dev)
    cg = torch.zeros(cs.n_params, device=dev)
    met = torch.zeros(8, device=dev)
    ctx.ppo_minibatch_fwd_bwd(_desc(ps), _t(pp, dev), pg, _desc(cs), _t(cp, dev), cg, met, _t(states, dev),
                              _t(actions, dev), _t(logp, dev), _t(returns, dev), _t(adv, dev), _t(idx, dev), hp)
    met = met.cpu().numpy()
    np.testing.assert_allclose(met[0], met_e["loss/policy_gradient_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(met[1], met_e["loss/critic_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(met[2], met_e["loss/entropy_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(met[3], met_e["policy_ratio/approx_kl"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(met[4], met_e["policy_ratio/clip_fraction"], rtol=1e-5, atol=1.5 / mb)
    loss = met[0] - ent * met[2] + cc * met[1]
    np.testing.assert_allclose(loss, loss_e, rtol=1e-5, atol=1e-6)
    # gradients: relative to each tensor's scale (1e-5 of the largest entry, fp32 accumulation noise)
    for got, exp, spec in ((pg.cpu().numpy(), gp_e, ps), (cg.cpu().numpy(), gc_e, cs)):
        segs = [(L["W"], L["in"] * L["out"]) for L in spec.layers] + [(spec.head["W"], spec.head["in"] * spec.head["out"])]
        for off, ln in segs:
            scale = np.abs(exp[off:off + ln]).max()
            np.testing.assert_allclose(got[off:off + ln], exp[off:off + ln], rtol=1e-4, atol=2e-5 * scale)
        np.testing.assert_allclose(got, exp, rtol=1e-3, atol=2e-5 * np.abs(exp).max())
        rel = np.linalg.norm(got - exp) / np.linalg.norm(exp)
        assert rel < 1e-5, rel


@pytest.mark.parametrize("arch", ["A", "B"])
def test_fused_and_unfused_first_layer_backward_agree(ctx, dev, arch):
    """k_dx_l1bwd (dX2 + LN' + dW1 fused, nothing of shape [mb, H1] in HBM) vs the unfused
    k_gemm_dx + k_l1<bwd> + k_gemm_dw_skinny path, mb not a multiple of the 32-row tile."""
    rng = np.random.default_rng(3)
    ps, pp, cs, cp, states, actions, logp, returns, adv, idx = _minibatch_case(arch, 17, 6, 3000, 1000, rng)
    hp = PpoHparams(0.1, 0.01, 0.7, 0.5, 0.9, 0.999, 1e-8)
    outs = []
    for disable in (0, 1):
        ctx.set_option("disable_l1fused", disable)
        pg = torch.zeros(ps.n_params, device=dev)
        cg = torch.zeros(cs.n_params, device=dev)
        met = torch.zeros(8, device=dev)
        ctx.ppo_minibatch_fwd_bwd(_desc(ps), _t(pp, dev), pg, _desc(cs), _t(cp, dev), cg, met, _t(states, dev),
                                  _t(actions, dev), _t(logp, dev), _t(returns, dev), _t(adv, dev), _t(idx, dev), hp)
        outs.append((pg.cpu().numpy(), cg.cpu().numpy()))
    ctx.set_option("disable_l1fused", 0)
    for a, b in zip(outs[0], outs[1]):
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-5


@pytest.mark.parametrize("arch,act", [("B", None), ("A", None), ("A", nets.ACT_RELU)])
def test_fused_first_layer_backward_equals_the_unfused_path(ctx, dev, arch, act):
    """k_dx_l1bwd (layer-2 input gradient + the whole first-layer backward per 32-row tile) against the unfused path
    (k_gemm_dx + k_l1<bwd> + k_gemm_dw_skinny) at a minibatch large enough that every workgroup walks several row tiles
    (20010 rows = 626 tiles over 256 workgroups, ragged last tile); reproducible bit for bit (fixed-order slabs)."""
    rng = np.random.default_rng(11)
    B, mb = 24000, 20010
    ps, pp, cs, cp, states, actions, logp, returns, adv, idx = _minibatch_case(arch, 17, 6, B, mb, rng)
    if act is not None:                       # the relu / 256 instantiation (SAC-shaped trunk) through the same entry point
        ps, cs = nets.MLPSpec(17, [256, 256], 6, act, False, True), nets.MLPSpec(17, [256, 256], 1, act, False, False)
        mean, _ = nets.forward(ps, pp, states)
        logp = (oppo.gaussian_log_prob(actions, mean, pp[ps.logstd:ps.logstd + 6][None, :]) + 0.05 * rng.standard_normal(B)).astype(np.float32)
    hp = PpoHparams(0.1, 0.01, 0.7, 0.5, 0.9, 0.999, 1e-8)
    dev_in = [_t(x, dev) for x in (states, actions, logp, returns, adv, idx)]
    P, C = _t(pp, dev), _t(cp, dev)
    outs = {}
    try:
        for name, unfused in (("fused", 0), ("fused2", 0), ("unfused", 1)):
            ctx.set_option("disable_l1fused", unfused)
            pg, cg, met = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.zeros(8, device=dev)
            ctx.ppo_minibatch_fwd_bwd(_desc(ps), P, pg, _desc(cs), C, cg, met, *dev_in, hp)
            torch.cuda.synchronize()
            outs[name] = (pg.cpu().numpy(), cg.cpu().numpy(), met.cpu().numpy())
    finally:
        ctx.set_option("disable_l1fused", 0)
    for k in range(3):
        assert np.array_equal(outs["fused"][k], outs["fused2"][k])
    for a, b in zip(outs["fused"][:2], outs["unfused"][:2]):
        assert np.isfinite(a).all()
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 1e-5
    np.testing.assert_array_equal(outs["fused"][2], outs["unfused"][2])            # the forward half is untouched


def test_first_layer_forward_mfma_equals_valu_kernel(ctx, dev):
    """k_l1fwd_mfma (K = 17 product on the matrix pipe, used from 1024 rows on) against k_l1<fwd> (VALU) on a ragged row
    count: same layer, different summation order."""
    rng = np.random.default_rng(5)
    ps, pp, cs, cp = _nets("B", 17, 6, rng)
    n = 3000 + 7
    x = _t(rng.standard_normal((n, 17)).astype(np.float32), dev)
    outs = []
    try:
        for on in (1, 0):
            ctx.set_option("l1fwd_mfma", on)
            o = torch.full((n, 6), float("nan"), device=dev)
            ctx.mlp_fwd(_desc(ps), _t(pp, dev), x, o)
            outs.append(o.cpu().numpy())
    finally:
        ctx.set_option("l1fwd_mfma", 1)
    assert np.isfinite(outs[0]).all()
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-5, atol=2e-6)


