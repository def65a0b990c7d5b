"""GPU: GAE scan, synthetic env step, clip+Adam vs the oracle (fp32, rel 1e-5)."""
import numpy as np
import pytest
import torch

from oracle import env as oenv
from oracle import nets
from oracle import ppo as oppo

pytestmark = pytest.mark.gpu
RTOL = 2e-5   # synthetic-env observations are Box-Muller draws on v_log_f32 / v_cos_f32: 1.1e-5 relative in the tails vs libm


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("T,N", [(1, 1), (5, 3), (7, 64), (128, 4096), (33, 1000)])
def test_gae_matches_oracle(ctx, dev, T, N):
    rng = np.random.default_rng(T * 1000 + N)
    r = rng.standard_normal((T, N)).astype(np.float32)
    v = rng.standard_normal((T, N)).astype(np.float32)
    nv = rng.standard_normal((T, N)).astype(np.float32)
    term = (rng.random((T, N)) < 0.1).astype(np.float32)
    adv_e, ret_e = oppo.gae(r, v, nv, term, 0.99, 0.9)
    adv = torch.empty(T, N, device=dev)
    ret = torch.empty(T, N, device=dev)
    ctx.gae(_t(r, dev), _t(v, dev), _t(nv, dev), _t(term, dev), adv, ret, 0.99, 0.9)
    np.testing.assert_allclose(adv.cpu().numpy(), adv_e, rtol=RTOL, atol=1e-5)
    np.testing.assert_allclose(ret.cpu().numpy(), ret_e, rtol=RTOL, atol=1e-5)


def test_gae_closed_form(ctx, dev):
    # no terminations, constant reward 1, zero values: A[t] = sum_k (gamma*lam)^k
    T, N = 16, 64
    z = torch.zeros(T, N, device=dev)
    adv = torch.empty(T, N, device=dev)
    ret = torch.empty(T, N, device=dev)
    ctx.gae(torch.ones(T, N, device=dev), z, z, z, adv, ret, 0.9, 0.5)
    gl = 0.45
    exp = np.array([(1 - gl ** (T - t)) / (1 - gl) for t in range(T)], dtype=np.float32)
    np.testing.assert_allclose(adv.cpu().numpy()[:, 0], exp, rtol=1e-6)
    # all terminated: A = r - v
    one = torch.ones(T, N, device=dev)
    ctx.gae(one * 2, one * 0.5, one * 7, one, adv, ret, 0.9, 0.5)
    np.testing.assert_allclose(adv.cpu().numpy(), 1.5, rtol=1e-6)
    np.testing.assert_allclose(ret.cpu().numpy(), 2.0, rtol=1e-6)


@pytest.mark.parametrize("N,O,A,off", [(64, 17, 6, 0), (130, 17, 6, 4096), (256, 4, 2, 0), (96, 33, 17, 7)])
def test_env_matches_oracle(ctx, dev, N, O, A, off):
    seed, horizon, p_term, noise = 5, 20, 0.05, 0.1
    o = oenv.RandomObsEnvOracle(seed, N, O, A, horizon, p_term, noise, off)
    obs_e = o.reset()
    obs = torch.empty(N, O, device=dev)
    ep_step = torch.empty(N, dtype=torch.int32, device=dev)
    ep_ret, last_ret, last_len = (torch.empty(N, device=dev) for _ in range(3))
    ctx.env_reset(seed, off, horizon, obs, ep_step, ep_ret, last_ret, last_len)
    np.testing.assert_allclose(obs.cpu().numpy(), obs_e, rtol=RTOL, atol=2e-6)
    assert np.array_equal(ep_step.cpu().numpy(), o.ep_step)
    fin = torch.empty(N, O, device=dev)
    rew, term, trunc = (torch.empty(N, device=dev) for _ in range(3))
    rng = np.random.default_rng(0)
    n_done = 0
    for t in range(45):
        a = rng.standard_normal((N, A)).astype(np.float32)
        obs_e, fin_e, r_e, term_e, trunc_e, done_e = o.step(a)
        ctx.env_step(seed, off, t, horizon, p_term, noise, _t(a, dev), obs, fin, rew, term, trunc, ep_step, ep_ret,
                     last_ret, last_len)
        assert np.array_equal(term.cpu().numpy() > 0.5, term_e)
        assert np.array_equal(trunc.cpu().numpy() > 0.5, trunc_e)
        np.testing.assert_allclose(rew.cpu().numpy(), r_e, rtol=RTOL, atol=1e-5)
        np.testing.assert_allclose(fin.cpu().numpy(), fin_e, rtol=RTOL, atol=2e-6)
        np.testing.assert_allclose(obs.cpu().numpy(), obs_e, rtol=RTOL, atol=2e-6)
        np.testing.assert_allclose(last_ret.cpu().numpy(), o.last_ret, rtol=1e-4, atol=1e-4)
        assert np.array_equal(ep_step.cpu().numpy(), o.ep_step)
        n_done += int(done_e.sum())
    assert n_done > 0  # resets and truncations were exercised


@pytest.mark.parametrize("n", [1, 5, 1000, 175244])
@pytest.mark.parametrize("max_norm", [0.5, 1e9, -1.0])
def test_clip_adam_matches_oracle(ctx, dev, n, max_norm):
    rng = np.random.default_rng(n)
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    pt, mt, vt = _t(p, dev), _t(m, dev), _t(v, dev)
    norm_out = torch.zeros(1, device=dev)
    for step in range(1, 4):
        g = (rng.standard_normal(n) * 0.01).astype(np.float32)
        gc, nrm = (g, oppo.global_norm(g)) if max_norm <= 0 else oppo.clip_by_global_norm(g, max_norm)
        p, m, v = oppo.adam_step(p, gc, m, v, step - 1, 3e-4)
        ctx.clip_adam_step(pt, _t(g, dev), mt, vt, step, 3e-4, max_norm, grad_norm_out=norm_out)
        np.testing.assert_allclose(norm_out.item(), nrm, rtol=1e-5)
        np.testing.assert_allclose(pt.cpu().numpy(), p, rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(mt.cpu().numpy(), m, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(vt.cpu().numpy(), v, rtol=RTOL, atol=1e-12)


def test_adam_first_step_identity(ctx, dev):
    # first Adam step with eps -> 0: delta = -lr * sign(g)
    g = torch.tensor([0.3, -2.0, 1e-3, -5e-2], device=dev)
    p = torch.zeros(4, device=dev)
    m, v = torch.zeros(4, device=dev), torch.zeros(4, device=dev)
    ctx.clip_adam_step(p, g, m, v, 1, 0.01, -1.0, eps=1e-20)
    np.testing.assert_allclose(p.cpu().numpy(), -0.01 * np.sign(g.cpu().numpy()), rtol=1e-5)


def test_grad_global_norm(ctx, dev):
    g = torch.randn(349837, device=dev)
    out = torch.zeros(1, device=dev)
    ctx.grad_global_norm(g, out)
    np.testing.assert_allclose(out.item(), float(torch.linalg.vector_norm(g.double())), rtol=1e-5)


@pytest.mark.parametrize("T,N,p_diff,arch", [(128, 4096, 0.002, "B"), (16, 70, 0.0, "B"), (7, 33, 1.0, "A"), (1, 50, 0.1, "B"),
                                            (12, 256, 0.3, "A")])
def test_next_values_reuse_equals_the_full_critic_pass(ctx, dev, T, N, p_diff, arch):
    """rlx_ppo_next_values_f32: values[t+1] where next_states[t] == states[t+1], the critic on the other rows (device-side
    row list, no host round trip) == the critic on all T*N rows of next_states, bit for bit."""
    from rlx_amd.hip import mlp_desc
    rng = np.random.default_rng(T * 1000 + N)
    O = 17
    cs = nets.make_spec(arch, O, 1, False)
    cp = (nets.init_params(cs, rng, 1.0) + 0.05 * rng.standard_normal(cs.n_params)).astype(np.float32)
    cd = mlp_desc(O, cs.hidden, 1, cs.act, cs.ln_first, False)
    states = rng.standard_normal((T, N, O)).astype(np.float32)
    next_states = np.empty_like(states)
    next_states[:-1] = states[1:]
    next_states[-1] = rng.standard_normal((N, O)).astype(np.float32)
    diff = rng.random((T, N)) < p_diff                      # episode ends: the final observation differs from the reset one
    next_states[diff] = rng.standard_normal((int(diff.sum()), O)).astype(np.float32)
    if T > 2:
        next_states[1, 0, 3] = -next_states[1, 0, 3] if next_states[1, 0, 3] != 0 else 1.0    # a single differing float
    C, S, NS = _t(cp, dev), _t(states, dev), _t(next_states, dev)
    # (bit-for-bit needs ONE engine on both sides: large-batch rlx_mlp_fwd_f32 calls use the fp16-pipe GEMMs by default, the
    #  compacted pass of rlx_ppo_next_values_f32 the exact-fp32 ones)
    ctx.set_option("gemm_bx", 0)
    try:
        values = torch.empty(T * N, 1, device=dev)
        ctx.mlp_fwd(cd, C, S.view(-1, O), values)
        full = torch.empty(T * N, 1, device=dev)
        ctx.mlp_fwd(cd, C, NS.view(-1, O), full)
        nv = torch.full((T, N), float("nan"), device=dev)
        ctx.ppo_next_values(cd, C, S, NS, values.view(T, N), nv)
        torch.cuda.synchronize()
        assert torch.equal(nv.view(-1), full.view(-1))
        nv2 = torch.full((T, N), float("nan"), device=dev)  # again: the row list's slot order may differ, the values may not
        ctx.ppo_next_values(cd, C, S, NS, values.view(T, N), nv2)
        assert torch.equal(nv, nv2)
    finally:
        ctx.set_option("gemm_bx", 1)


def test_reduce_metrics_matches_the_reference_formulas(ctx, dev):
    """rlx_ppo_reduce_metrics_f32: column means of the per-update metric rows, explained variance
    1 - var(returns - values) / (var(returns) + 1e-8) and mean(exp(logstd)) (ppo/flax/ppo.py:215-216, 226-230, 300-307)
    against numpy float64; ragged size, reproducible bit for bit."""
    rng = np.random.default_rng(21)
    n_upd, n, A = 37, 128 * 4096 + 13, 6
    met = rng.standard_normal((n_upd, 10)).astype(np.float32)
    ret = (3.0 + 2.0 * rng.standard_normal(n)).astype(np.float32)
    val = (ret + 0.5 * rng.standard_normal(n)).astype(np.float32)
    ls = (0.3 * rng.standard_normal(A)).astype(np.float32)
    outs = []
    for _ in range(2):
        out = torch.full((12,), float("nan"), device=dev)
        ctx.ppo_reduce_metrics(_t(met, dev), _t(ret, dev), _t(val, dev), _t(ls, dev), out)
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    r64, d64 = ret.astype(np.float64), ret.astype(np.float64) - val.astype(np.float64)
    exp = np.concatenate([met.astype(np.float64).mean(0), [1.0 - d64.var() / (r64.var() + 1e-8)], [np.exp(ls.astype(np.float64)).mean()]])
    np.testing.assert_allclose(outs[0], exp, rtol=2e-6, atol=1e-7)
    out = torch.full((12,), float("nan"), device=dev)
    ctx.ppo_reduce_metrics(_t(met, dev), _t(ret, dev), _t(val, dev), None, out)      # Categorical policy: no logstd
    assert out[11].item() == 0.0
