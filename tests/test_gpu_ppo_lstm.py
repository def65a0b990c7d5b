"""GPU: recurrent PPO (rlx_ppo_lstm_*) against oracle/ppo_lstm.py (torch autograd, float64).
fp32 kernels; tolerances: 1e-5 relative on forward quantities and the critic, 2e-5 on the policy-gradient loss, 5e-5 of each
parameter block's gradient scale on BPTT gradients (per-element error of a sum over T*n rows of fp32 products that has a
different association than the float64 oracle; the relative error of the whole gradient VECTOR is below 1e-5)."""
import numpy as np
import pytest
import torch

from oracle import nets, ppo as oppo, ppo_lstm as ol, prng
from rlx_amd.hip import PpoHparams, mlp_desc
from rlx_amd.hip.lib import lstm_policy_desc

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _hp(clip=0.2, ent=0.01, vf=0.5, mgn=0.5):
    hp = PpoHparams()
    hp.clip_range, hp.entropy_coef, hp.critic_coef, hp.max_grad_norm = clip, ent, vf, mgn
    hp.adam_b1, hp.adam_b2, hp.adam_eps = 0.9, 0.999, 1e-5
    return hp


def _setup(O, A, rng, share=False, perturb=0.03, cell="lstm", combine="concat"):
    spec = ol.LstmPolicySpec(O, A, 128, 64, (512, 256, 128), share, cell, combine)
    p = ol.init_params(spec, rng, 1.0)
    p = (p + perturb * rng.standard_normal(p.shape)).astype(np.float32)
    cs = nets.make_spec("B", O, 1, False)
    cp = nets.init_params(cs, rng, 1.0)
    cp = (cp + perturb * rng.standard_normal(cp.shape)).astype(np.float32)
    return spec, p, cs, cp


def _ldesc(spec):
    return lstm_policy_desc(spec.O, spec.A, spec.E, spec.H, spec.torso, spec.share, 1 if spec.cell == "gru" else 0,
                            1 if spec.combine == "film" else 0)


def _cdesc(cs):
    return mlp_desc(cs.in_dim, cs.hidden, cs.out_dim, cs.act, cs.ln_first, cs.has_logstd)


def test_param_count(ctx):
    for cell in ("lstm", "gru"):
        for share in (False, True):
            for combine in ("concat", "film"):
                spec = ol.LstmPolicySpec(17, 6, 128, 64, (512, 256, 128), share, cell, combine)
                assert ctx.lstm_policy_param_count(_ldesc(spec)) == spec.n_params


@pytest.mark.parametrize("fused", [1, 0, 2], ids=["fused", "separate", "fused-fp16-pipe-layers"])
@pytest.mark.parametrize("cell", ["lstm", "gru"])
@pytest.mark.parametrize("n", [1, 32, 70])
def test_act_matches_oracle(ctx, dev, n, cell, fused):
    ctx.set_option("fused_recurrent_act", 1 if fused else 0)
    try:
        _act_case(ctx, dev, n, cell, images=(fused == 2))
    finally:
        ctx.set_option("fused_recurrent_act", 1)
        ctx.rollout_end()


def _act_case(ctx, dev, n, cell, images=False, critic_cols=None):
    rng = np.random.default_rng(n)
    O, A = 17, 6
    spec, p, cs, cp = _setup(O, A, rng, cell=cell)
    obs = rng.standard_normal((n, O)).astype(np.float32)
    cobs = obs
    if critic_cols is not None:     # the critic reads its own observation columns (critic_observation_indices): another width
        cs = nets.make_spec("B", critic_cols, 1, False)
        cp = (nets.init_params(cs, rng, 1.0) + 0.03 * rng.standard_normal(cs.n_params)).astype(np.float32)
        cobs = rng.standard_normal((n, critic_cols)).astype(np.float32)
    c = (0.5 * rng.standard_normal((n, 64))).astype(np.float32)
    h = np.tanh(0.5 * rng.standard_normal((n, 64))).astype(np.float32)
    key = prng.prng_key(11 + n)
    t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    mean, c2, h2 = ol.apply_one_step(spec, t64(p), t64(obs), t64(c), t64(h))
    ks = prng.split(key, 2)
    eps = prng.normal(ks[1], (n, A))
    logstd = p[spec.off["logstd"][0]:][:A].astype(np.float64)
    act = mean.numpy() + np.exp(logstd) * eps
    logp = oppo.gaussian_log_prob(act, mean.numpy(), logstd[None, :])
    val, _ = nets.forward(cs, cp.astype(np.float64), cobs.astype(np.float64))
    cd, hd = _t(c, dev), _t(h, dev)
    action = torch.empty(n, A, device=dev)
    proc = torch.empty(n, A, device=dev)
    value = torch.empty(n, device=dev)
    lp = torch.empty(n, device=dev)
    lo, hi = _t(np.full(A, -2.0, np.float32), dev), _t(np.full(A, 3.0, np.float32), dev)
    Pd, Cd = _t(p, dev), _t(cp, dev)
    if images:      # the decoder's hidden layers on the fp16 pipe from images laid out once per rollout
        ctx.ppo_lstm_rollout_begin(_ldesc(spec), Pd, _cdesc(cs), Cd)
    k2 = ctx.ppo_lstm_act(_ldesc(spec), Pd, _cdesc(cs), Cd, _t(obs, dev), cd, hd, key, action, proc, value, lp,
                          clip_and_rescale=True, act_low=lo, act_high=hi,
                          critic_obs=None if critic_cols is None else _t(cobs, dev))
    assert np.array_equal(k2, ks[0])
    np.testing.assert_allclose(cd.cpu().numpy(), c2.numpy(), rtol=1e-5, atol=2e-6)   # GRU: untouched
    np.testing.assert_allclose(hd.cpu().numpy(), h2.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(action.cpu().numpy(), act, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(lp.cpu().numpy(), logp, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(value.cpu().numpy(), val[:, 0], rtol=1e-5, atol=5e-6)
    exp_proc = -2.0 + 0.5 * (np.clip(act, -1, 1) + 1.0) * 5.0
    np.testing.assert_allclose(proc.cpu().numpy(), exp_proc, rtol=1e-5, atol=1e-5)


def test_mask_carry(ctx, dev):
    rng = np.random.default_rng(0)
    n = 50
    c, h = rng.standard_normal((n, 64)).astype(np.float32), rng.standard_normal((n, 64)).astype(np.float32)
    term = (rng.random(n) < 0.3).astype(np.float32)
    trunc = (rng.random(n) < 0.3).astype(np.float32)
    done = np.maximum(term, trunc)
    cd, hd, dd = _t(c, dev), _t(h, dev), torch.empty(n, device=dev)
    ctx.lstm_mask_carry(cd, hd, _t(term, dev), _t(trunc, dev), dd)
    assert np.array_equal(dd.cpu().numpy(), done)
    assert np.array_equal(cd.cpu().numpy(), c * (1 - done)[:, None])
    assert np.array_equal(hd.cpu().numpy(), h * (1 - done)[:, None])


@pytest.mark.parametrize("fused", [1, 0, 2], ids=["fused", "separate", "fused-fp16-pipe-layers"])
def test_act_with_the_critics_own_observation_columns(ctx, dev, fused):
    """critic_observation_indices (ppo_lstm/flax_full_jit/critic.py:12,23): the value comes from the critic's rows, the
    action / log-prob / carry from the policy's."""
    ctx.set_option("fused_recurrent_act", 1 if fused else 0)
    try:
        _act_case(ctx, dev, 33, "lstm", images=(fused == 2), critic_cols=23)
    finally:
        ctx.set_option("fused_recurrent_act", 1)
        ctx.rollout_end()
    with pytest.raises(Exception, match="critic_obs"):        # widths differ and no critic rows: refused, not mis-read
        rng = np.random.default_rng(0)
        spec, p, _, _ = _setup(17, 6, rng)
        cs = nets.make_spec("B", 23, 1, False)
        z = lambda *sh: torch.zeros(*sh, device=dev)
        ctx.ppo_lstm_act(_ldesc(spec), _t(p, dev), _cdesc(cs), z(cs.n_params), z(4, 17), z(4, 64), z(4, 64), prng.prng_key(1),
                         z(4, 6), z(4, 6), z(4), z(4))


def _rollout_case(spec, p, T, N, rng, p_done=0.15):
    O, A = spec.O, spec.A
    states = rng.standard_normal((T, N, O)).astype(np.float32)
    actions = rng.standard_normal((T, N, A)).astype(np.float32)
    dones = (rng.random((T, N)) < p_done).astype(np.float32)
    c0 = (0.5 * rng.standard_normal((N, spec.H))).astype(np.float32)
    h0 = np.tanh(0.5 * rng.standard_normal((N, spec.H))).astype(np.float32)
    t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    with torch.no_grad():
        mean = ol.forward_sequence(spec, t64(p), t64(states), t64(dones), t64(c0), t64(h0)).numpy()
    logstd = p[spec.off["logstd"][0]:][:A].astype(np.float64)[None, None, :]
    logp = ((-0.5 * ((actions - mean) / np.exp(logstd)) ** 2 - 0.5 * ol.LOG_2PI - logstd).sum(-1)
            + 0.05 * rng.standard_normal((T, N))).astype(np.float32)
    returns = rng.standard_normal((T, N)).astype(np.float32)
    adv = (rng.standard_normal((T, N)) * 2 + 0.3).astype(np.float32)
    return states, actions, logp, returns, adv, dones, c0, h0


def _oracle_grads(spec, p, cs, cp, case, env_idx, hp):
    states, actions, logp, returns, adv, dones, c0, h0 = case
    t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    P = t64(p).requires_grad_(True)
    C = t64(cp).requires_grad_(True)
    a = adv[:, env_idx].astype(np.float64)
    a = (a - a.mean()) / (a.std() + 1e-8)
    loss, met = ol.ppo_lstm_loss(spec, P, cs, C, t64(states[:, env_idx]), t64(actions[:, env_idx]), t64(logp[:, env_idx]),
                                 t64(returns[:, env_idx]), t64(a), t64(dones[:, env_idx]), t64(c0[env_idx]), t64(h0[env_idx]),
                                 hp.clip_range, hp.entropy_coef, hp.critic_coef)
    loss.backward()
    return {k: float(v) for k, v in met.items()}, P.grad.numpy(), C.grad.numpy()


@pytest.mark.parametrize("cell", ["lstm", "gru"])
@pytest.mark.parametrize("T,N,ne,share", [(8, 48, 40, False), (5, 32, 32, False), (16, 70, 33, True), (3, 4, 1, False)])
def test_minibatch_grads_match_autograd(ctx, dev, T, N, ne, share, cell):
    rng = np.random.default_rng(T * 1000 + N)
    spec, p, cs, cp = _setup(17, 6, rng, share, cell=cell)
    case = _rollout_case(spec, p, T, N, rng)
    env_idx = rng.permutation(N)[:ne].astype(np.int32)
    hp = _hp()
    met_o, gp_o, gc_o = _oracle_grads(spec, p, cs, cp, case, env_idx, hp)
    pg = torch.empty(spec.n_params, device=dev)
    cg = torch.empty(cs.n_params, device=dev)
    met = torch.empty(8, device=dev)
    ctx.ppo_lstm_minibatch_fwd_bwd(_ldesc(spec), _t(p, dev), pg, _cdesc(cs), _t(cp, dev), cg, met, *[_t(x, dev) for x in case],
                                   _t(env_idx, dev), hp)
    m = met.cpu().numpy()
    assert m[0] == pytest.approx(met_o["loss/policy_gradient_loss"], rel=1e-5, abs=1e-6)   # mean of signed O(1) terms: 1e-6 absolute
    assert m[1] == pytest.approx(met_o["loss/critic_loss"], rel=1e-5)
    assert m[2] == pytest.approx(met_o["loss/entropy_loss"], rel=1e-5)
    assert m[3] == pytest.approx(met_o["policy_ratio/approx_kl"], rel=1e-4, abs=2e-8)
    assert m[4] == pytest.approx(met_o["policy_ratio/clip_fraction"], abs=2.0 / (T * ne))
    gp, gc = pg.cpu().numpy().astype(np.float64), cg.cpu().numpy().astype(np.float64)
    # per parameter block: error relative to that block's gradient scale
    for name, (o, n) in spec.off.items():
        ref = gp_o[o:o + n]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(gp[o:o + n] - ref).max()
        assert err <= 5e-5 * scale + 1e-7, (name, err, scale)          # worst single ENTRY of the block (max norm)
        assert np.linalg.norm(gp[o:o + n] - ref) <= 1e-5 * np.linalg.norm(ref) + 1e-9, (name, "L2")   # the 1e-5 bar: L2-relative
    assert np.linalg.norm(gp - gp_o) / np.linalg.norm(gp_o) < 1e-5
    assert np.abs(gc - gc_o).max() <= 2e-5 * np.abs(gc_o).max()
    assert np.linalg.norm(gc - gc_o) / np.linalg.norm(gc_o) < 1e-5
    print(f"recurrent minibatch ({cell}, T={T}, ne={ne}): ||dg||/||g|| policy {np.linalg.norm(gp - gp_o) / np.linalg.norm(gp_o):.2e} "
          f"critic {np.linalg.norm(gc - gc_o) / np.linalg.norm(gc_o):.2e}, pg loss abs err {abs(m[0] - met_o['loss/policy_gradient_loss']):.2e}")


@pytest.mark.parametrize("cell", ["lstm", "gru"])
def test_minibatch_with_observation_index_sets(ctx, dev, cell):
    """policy_observation_indices / critic_observation_indices: `states` holds the policy's columns, hparams.critic_states
    the critic's.  The loss separates, so the oracle runs twice -- the policy on its columns, the critic on its own."""
    rng = np.random.default_rng(77)
    T, N, ne, O, A = 6, 40, 24, 24, 5
    pidx, cidx = np.arange(0, 17), np.arange(5, 24)
    spec, p, _, _ = _setup(len(pidx), A, rng, cell=cell)
    cs = nets.make_spec("B", len(cidx), 1, False)
    cp = (nets.init_params(cs, rng, 1.0) + 0.03 * rng.standard_normal(cs.n_params)).astype(np.float32)
    case = list(_rollout_case(spec, p, T, N, rng))
    full = rng.standard_normal((T, N, O)).astype(np.float32)
    full[..., pidx] = case[0]                                        # the env's rows; the policy's columns are the case's states
    crit = np.ascontiguousarray(full[..., cidx])
    env_idx = rng.permutation(N)[:ne].astype(np.int32)
    hp = _hp()
    # oracle, policy: critic of the policy's width (its gradient is discarded); oracle, critic: a policy on the critic's rows
    cs_p = nets.make_spec("B", len(pidx), 1, False)
    met_p, gp_o, _ = _oracle_grads(spec, p, cs_p, nets.init_params(cs_p, rng, 1.0), tuple(case), env_idx, hp)
    spec_c, p_c, _, _ = _setup(len(cidx), A, rng, cell=cell)
    met_c, _, gc_o = _oracle_grads(spec_c, p_c, cs, cp, (crit,) + tuple(case[1:]), env_idx, hp)
    pg, cg, met = torch.empty(spec.n_params, device=dev), torch.empty(cs.n_params, device=dev), torch.empty(8, device=dev)
    crit_d = _t(crit, dev)
    hp.critic_states = crit_d.data_ptr()
    ctx.ppo_lstm_minibatch_fwd_bwd(_ldesc(spec), _t(p, dev), pg, _cdesc(cs), _t(cp, dev), cg, met, *[_t(x, dev) for x in case],
                                   _t(env_idx, dev), hp)
    m = met.cpu().numpy()
    assert m[0] == pytest.approx(met_p["loss/policy_gradient_loss"], rel=1e-5, abs=1e-6)
    assert m[1] == pytest.approx(met_c["loss/critic_loss"], rel=1e-5)
    gp, gc = pg.cpu().numpy().astype(np.float64), cg.cpu().numpy().astype(np.float64)
    assert np.linalg.norm(gp - gp_o) / np.linalg.norm(gp_o) < 1e-5
    assert np.linalg.norm(gc - gc_o) / np.linalg.norm(gc_o) < 1e-5


@pytest.mark.parametrize("cell", ["lstm", "gru"])
def test_update_matches_oracle_loop(ctx, dev, cell):
    """rlx_ppo_lstm_update_f32 == env-index permutation (oracle prng) + per-minibatch autograd + oracle clip/Adam."""
    rng = np.random.default_rng(5)
    T, N, E, mbs = 4, 64, 2, 4 * 32
    ne, M = mbs // T, N // (mbs // T)
    spec, p, cs, cp = _setup(17, 6, rng, cell=cell)
    case = _rollout_case(spec, p, T, N, rng)
    hp = _hp()
    key = prng.prng_key(9)
    k_exp, idx = ol.env_minibatch_indices(key, N, E, M, ne)
    lr = np.linspace(3e-4, 1e-4, E * M).astype(np.float32)
    # oracle loop (float64 gradients, float32 optimizer state like the device)
    po, co = p.copy(), cp.copy()
    pm, pv, cm, cv = (np.zeros_like(x) for x in (p, p, cp, cp))
    for u in range(E * M):
        _, gp, gc = _oracle_grads(spec, po, cs, co, case, idx[u], hp)
        gp, _ = oppo.clip_by_global_norm(gp.astype(np.float32), hp.max_grad_norm)
        gc, _ = oppo.clip_by_global_norm(gc.astype(np.float32), hp.max_grad_norm)
        po, pm, pv = oppo.adam_step(po, gp, pm, pv, u, lr[u], eps=1e-5)
        co, cm, cv = oppo.adam_step(co, gc, cm, cv, u, lr[u], eps=1e-5)
    pd, cd = _t(p, dev), _t(cp, dev)
    z = lambda n: torch.zeros(n, device=dev)
    met = torch.empty(E * M, 10, device=dev)
    k2, cnt = ctx.ppo_lstm_update(_ldesc(spec), pd, z(spec.n_params), z(spec.n_params), _cdesc(cs), cd, z(cs.n_params),
                                  z(cs.n_params), *[_t(x, dev) for x in case], E, mbs, key, 0, lr, hp, met)
    assert cnt == E * M
    assert np.array_equal(k2, k_exp)
    assert np.isfinite(met.cpu().numpy()).all()
    # Adam's first steps move every parameter by ~lr whatever the gradient size, so a parameter whose gradient is
    # rounding noise may step the other way: almost all within 2e-5, none further than 2 * lr * steps
    for got, exp in ((pd.cpu().numpy(), po), (cd.cpu().numpy(), co)):
        d = np.abs(got - exp)
        assert (d <= 2e-5 + 1e-3 * np.abs(exp)).mean() > 0.999, (d.max(), (d > 2e-5).mean())
        assert d.max() <= 2 * 3e-4 * cnt


def test_golden_ppo_lstm_fixture(ctx, dev):
    """HIP vs the committed golden vectors (tests/golden/ppo_lstm.npz): loss terms, BPTT gradients, env-index permutation."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ppo_lstm.npz"))
    O, A = int(g["obs_dim"]), int(g["act_dim"])
    spec = ol.LstmPolicySpec(O, A, 128, 64, (512, 256, 128), False)
    cs = nets.make_spec("B", O, 1, False)
    hp = _hp(float(g["clip_range"]), float(g["entropy_coef"]), float(g["critic_coef"]))
    pg, cg, met = torch.empty(spec.n_params, device=dev), torch.empty(cs.n_params, device=dev), torch.empty(8, device=dev)
    case = [_t(g[k], dev) for k in ("states", "actions", "log_probs", "returns", "advantages", "dones", "c0", "h0")]
    ctx.ppo_lstm_minibatch_fwd_bwd(_ldesc(spec), _t(g["pparams"], dev), pg, _cdesc(cs), _t(g["cparams"], dev), cg, met, *case,
                                   _t(g["env_idx"], dev), hp)
    m = met.cpu().numpy()
    assert m[0] == pytest.approx(float(g["pg_loss"]), rel=2e-5, abs=2e-6)
    assert m[1] == pytest.approx(float(g["critic_loss"]), rel=1e-5)
    assert m[3] == pytest.approx(float(g["approx_kl"]), rel=1e-4, abs=2e-8)
    for name, (o, n) in spec.off.items():
        ref = g["pgrads"][o:o + n].astype(np.float64)
        err = np.abs(pg.cpu().numpy()[o:o + n] - ref).max()
        assert err <= 5e-5 * max(np.abs(ref).max(), 1e-6) + 1e-7, (name, err)
    assert np.abs(cg.cpu().numpy() - g["cgrads"]).max() <= 2e-5 * np.abs(g["cgrads"]).max()
    # env-index permutation of the whole-update entry point: bit-exact through the key chain
    N = g["states"].shape[1]
    perm = torch.empty(2 * N, dtype=torch.int32, device=dev)
    k2 = ctx.permutation(g["key"], perm, 2, N)
    assert np.array_equal(k2, g["perm_key"])
    assert np.array_equal(perm.cpu().numpy().reshape(-1, 8), g["perm_env_idx"])


@pytest.mark.parametrize("cell", ["lstm", "gru"])
@pytest.mark.parametrize("T,N,ne,share", [(8, 48, 40, False), (16, 70, 33, True), (3, 4, 1, False)])
def test_film_minibatch_grads_match_autograd(ctx, dev, T, N, ne, share, cell):
    """lstm_obs_combine_method = "film" (ppo_lstm/flax_full_jit/policy.py:55-57,95-100): gamma / beta from the cell latent,
    torso input = obs_latent * gamma + beta.  Acting step and sequence minibatch (loss + BPTT gradients) vs float64 autograd."""
    rng = np.random.default_rng(T * 77 + N)
    spec, p, cs, cp = _setup(17, 6, rng, share, cell=cell, combine="film")
    assert spec.K1 == 128 and "film.W" in spec.off
    # acting step
    n = 33
    obs = rng.standard_normal((n, 17)).astype(np.float32)
    c = (0.5 * rng.standard_normal((n, 64))).astype(np.float32)
    h = np.tanh(0.5 * rng.standard_normal((n, 64))).astype(np.float32)
    t64 = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    mean, c2, h2 = ol.apply_one_step(spec, t64(p), t64(obs), t64(c), t64(h))
    cd, hd = _t(c, dev), _t(h, dev)
    action, proc = torch.empty(n, 6, device=dev), torch.empty(n, 6, device=dev)
    value, lp = torch.empty(n, device=dev), torch.empty(n, device=dev)
    ctx.ppo_lstm_act(_ldesc(spec), _t(p, dev), _cdesc(cs), _t(cp, dev), _t(obs, dev), cd, hd, prng.prng_key(3), action, proc, value,
                     lp, deterministic=True)
    np.testing.assert_allclose(action.cpu().numpy(), mean.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(hd.cpu().numpy(), h2.numpy(), rtol=1e-5, atol=2e-6)
    # sequence minibatch
    case = _rollout_case(spec, p, T, N, rng)
    env_idx = rng.permutation(N)[:ne].astype(np.int32)
    hp = _hp()
    met_o, gp_o, gc_o = _oracle_grads(spec, p, cs, cp, case, env_idx, hp)
    pg, cg, met = torch.empty(spec.n_params, device=dev), torch.empty(cs.n_params, device=dev), torch.empty(8, device=dev)
    ctx.ppo_lstm_minibatch_fwd_bwd(_ldesc(spec), _t(p, dev), pg, _cdesc(cs), _t(cp, dev), cg, met, *[_t(x, dev) for x in case],
                                   _t(env_idx, dev), hp)
    m = met.cpu().numpy()
    assert m[0] == pytest.approx(met_o["loss/policy_gradient_loss"], rel=1e-5, abs=1e-6)   # mean of signed O(1) terms: 1e-6 absolute
    assert m[3] == pytest.approx(met_o["policy_ratio/approx_kl"], rel=1e-4, abs=2e-8)
    gp = pg.cpu().numpy().astype(np.float64)
    for name, (o, nn_) in spec.off.items():
        ref = gp_o[o:o + nn_]
        scale = max(np.abs(ref).max(), 1e-6)
        err = np.abs(gp[o:o + nn_] - ref).max()
        assert err <= 5e-5 * scale + 1e-7, (name, err, scale)
