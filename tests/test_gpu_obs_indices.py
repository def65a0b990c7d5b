"""GPU: networks that read a SUBSET of the env's observation -- `x[..., self.policy_observation_indices]`
(rl_x/algorithms/ppo/flax/policy.py:13,33; critic.py:12,24; full-jit policy.py:15,32, critic.py:12,23; sac/flax/policy.py:14,31,
critic.py:11,23; env side e.g. rl_x/environments/mujoco_playground/g1_joystick_flat_terrain/mjx/wrappers.py:22-23).
The column selection is rlx_select_columns_f32; the PPO / SAC updates take the critic's own rows through
rlx_ppo_hparams.critic_states / rlx_sac_hparams.critic_states.  Checked against the oracle evaluated on the selected columns
(disjoint index sets, so a net reading the wrong columns cannot pass) and through the plugins on an env that defines the sets."""
import numpy as np
import pytest
import torch

from oracle import nets, ppo as oppo, prng, sac as osac
from rlx_amd.hip import PpoHparams, SacHparams, mlp_desc
from rlx_amd.hip import lib as L

pytestmark = pytest.mark.gpu


def _t(a, dev, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(dev)


def _desc(spec):
    return mlp_desc(spec.in_dim, spec.hidden, spec.out_dim, spec.act, spec.ln_first, spec.has_logstd)


def test_select_columns(ctx, dev):
    rng = np.random.default_rng(0)
    for M, O, n, ldo in ((1, 5, 3, 3), (130, 52, 17, 20), (4096, 376, 200, 200), (7, 3, 1, 4)):
        x = rng.standard_normal((M, O)).astype(np.float32)
        cols = rng.permutation(O)[:n].astype(np.int32)
        out = torch.full((M, ldo), -7.0, device=dev)
        ctx.select_columns(_t(x, dev), _t(cols, dev, np.int32), out)
        got = out.cpu().numpy()
        assert np.array_equal(got[:, :n], x[:, cols]) and (got[:, n:] == -7.0).all()


@pytest.mark.parametrize("arch,O,Op,Oc,B,mb", [("B", 52, 12, 40, 1024, 256), ("A", 30, 13, 17, 700, 130), ("B", 60, 17, 40, 8192, 4096)])
def test_ppo_minibatch_with_disjoint_policy_and_critic_columns(ctx, dev, arch, O, Op, Oc, B, mb):
    rng = np.random.default_rng(O + mb)
    perm = rng.permutation(O)
    pidx, cidx = np.sort(perm[:Op]).astype(np.int32), np.sort(perm[Op:Op + Oc]).astype(np.int32)     # disjoint
    A = 6
    ps, cs = nets.make_spec(arch, Op, A, True), nets.make_spec(arch, Oc, 1, False)
    pp = (nets.init_params(ps, rng, 0.01) + 0.05 * rng.standard_normal(ps.n_params)).astype(np.float32)
    cp = (nets.init_params(cs, rng, 1.0) + 0.05 * rng.standard_normal(cs.n_params)).astype(np.float32)
    full = rng.standard_normal((B, O)).astype(np.float32)
    actions = rng.standard_normal((B, A)).astype(np.float32)
    returns = rng.standard_normal(B).astype(np.float32)
    adv = (rng.standard_normal(B) * 2 + 0.3).astype(np.float32)
    idx = rng.permutation(B)[:mb].astype(np.int32)
    # the rows the update reads: selected ON THE DEVICE from the full observation rows
    sp, sc = torch.empty(B, Op, device=dev), torch.empty(B, Oc, device=dev)
    ctx.select_columns(_t(full, dev), _t(pidx, dev, np.int32), sp)
    ctx.select_columns(_t(full, dev), _t(cidx, dev, np.int32), sc)
    mean, _ = nets.forward(ps, pp, full[:, pidx])
    logp = (oppo.gaussian_log_prob(actions, mean, pp[ps.logstd:ps.logstd + A][None, :]) + 0.05 * rng.standard_normal(B)).astype(np.float32)
    clip, ent, cc = 0.1, 0.01, 0.7
    f64 = lambda a: a.astype(np.float64)
    madv = oppo.normalize_advantages(f64(adv[idx]))
    loss_e, met_e, gp_e, gc_e = oppo.ppo_loss_and_grads(ps, f64(pp), cs, f64(cp), f64(full[idx][:, pidx]), f64(actions[idx]),
                                                        f64(logp[idx]), f64(returns[idx]), madv, clip, ent, cc,
                                                        critic_states=f64(full[idx][:, cidx]))
    hp = PpoHparams(clip, ent, cc, 0.5, 0.9, 0.999, 1e-8)
    hp.critic_states = sc.data_ptr()
    pg, cg, met = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.zeros(8, device=dev)
    ctx.ppo_minibatch_fwd_bwd(_desc(ps), _t(pp, dev), pg, _desc(cs), _t(cp, dev), cg, met, sp, _t(actions, dev), _t(logp, dev),
                              _t(returns, dev), _t(adv, dev), _t(idx, dev, np.int32), hp)
    m = met.cpu().numpy()
    np.testing.assert_allclose(m[0], met_e["loss/policy_gradient_loss"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(m[1], met_e["loss/critic_loss"], rtol=1e-5, atol=1e-6)
    assert np.linalg.norm(pg.cpu().numpy() - gp_e) / np.linalg.norm(gp_e) < 1e-5
    assert np.linalg.norm(cg.cpu().numpy() - gc_e) / np.linalg.norm(gc_e) < 1e-5
    # a critic fed the POLICY's columns must not pass (same width on purpose in the ("A", 30, 13, 17) case would not even run)
    if Op == Oc:
        hp.critic_states = sp.data_ptr()
        ctx.ppo_minibatch_fwd_bwd(_desc(ps), _t(pp, dev), pg, _desc(cs), _t(cp, dev), cg, met, sp, _t(actions, dev), _t(logp, dev),
                                  _t(returns, dev), _t(adv, dev), _t(idx, dev, np.int32), hp)
        assert np.linalg.norm(cg.cpu().numpy() - gc_e) / np.linalg.norm(gc_e) > 1e-2


def test_whole_update_with_critic_rows_equal_to_the_shared_rows_is_bit_identical(ctx, dev):
    """critic_states pointing at a COPY of the shared observation rows goes through the separate gather (mb_xc) of the
    pipelined two-chain update: same numbers in, so parameters, moments, metrics and key must come out bit for bit."""
    T, N, O, A, mb, E = 16, 512, 17, 6, 2048, 2
    rng = np.random.default_rng(4)
    ps, cs = nets.make_spec("B", O, A, True), nets.make_spec("B", O, 1, False)
    pp = (nets.init_params(ps, rng, 0.01) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    cp = (nets.init_params(cs, rng, 1.0) + 0.02 * rng.standard_normal(cs.n_params)).astype(np.float32)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    states, actions, logp, returns, adv = r(T, N, O), r(T, N, A), 0.1 * r(T, N) - 8.5, r(T, N), 2 * r(T, N) + 0.5
    n_upd = E * (T * N // mb)
    lr = np.full(n_upd, 4e-4, np.float32)
    outs = []
    for own_rows in (False, True):
        hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)
        crows = states.clone()
        if own_rows:
            hp.critic_states = crows.data_ptr()
        P, C = _t(pp, dev), _t(cp, dev)
        z = lambda x: torch.zeros_like(x)
        mom = [z(P), z(P), z(C), z(C)]
        met = torch.empty(n_upd, 10, device=dev)
        key, cnt = ctx.ppo_update(_desc(ps), P, mom[0], mom[1], _desc(cs), C, mom[2], mom[3], states, actions, logp, returns, adv,
                                  E, mb, L.prng_key(9), 0, lr, hp, met)
        torch.cuda.synchronize()
        outs.append([P, C, met] + mom + [torch.from_numpy(np.asarray(key).astype(np.int64))])
    for a, b in zip(*outs):
        assert torch.equal(a.cpu(), b.cpu())
    assert bool(torch.isfinite(outs[0][2]).all())


@pytest.mark.parametrize("arch,O,Op,Oc,B", [("flax", 60, 20, 33, 256), ("flax", 400, 376, 24, 4096), ("full_jit", 50, 11, 37, 200),
                                            ("flax", 45, 41, 3, 130)])
def test_sac_update_with_disjoint_policy_and_critic_columns(ctx, dev, arch, O, Op, Oc, B):
    """Policy on Op columns, twin critics on Oc OTHER columns (+ the action): narrow / wide / ragged widths on both sides
    (Op = 41: a wide policy observation whose width is not a multiple of 4 takes the padded copy)."""
    rng = np.random.default_rng(O + B)
    A, H = 5, 64
    perm = rng.permutation(O)
    pidx, cidx = np.sort(perm[:Op]).astype(np.int32), np.sort(perm[Op:Op + Oc]).astype(np.int32)
    ps, _ = osac.make_specs(Op, A, H, arch=arch)
    _, qs = osac.make_specs(Oc, A, H, arch=arch)
    pp = (osac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    pp[ps.head["W"]:ps.head["W"] + ps.head["in"] * ps.head["out"]] *= 0.1
    qp = (np.concatenate([osac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    qtp = (qp + 0.01 * rng.standard_normal(qp.shape)).astype(np.float32)
    s, s2 = rng.standard_normal((B, O)).astype(np.float32), rng.standard_normal((B, O)).astype(np.float32)
    a = np.tanh(rng.standard_normal((B, A))).astype(np.float32)
    r = rng.standard_normal(B).astype(np.float32)
    term = (rng.random(B) < 0.2).astype(np.float32)
    log_alpha, gamma, tau, lr = np.float32(-0.3), 0.99, 0.005, 3e-4
    key = prng.prng_key(21)
    sched = int(arch == "full_jit")
    f = lambda x: x.astype(np.float64)
    new_key_e, e1, e2 = osac.sample_noise(key, B, A, True, schedule=sched)
    met_e, gp_e, gq_e, ga_e = osac.loss_and_grads(ps, f(pp), qs, f(qp), f(qtp), np.float64(log_alpha), f(s[:, pidx]), f(s2[:, pidx]),
                                                 f(a), f(r), f(term), f(e1), f(e2), gamma, -float(A),
                                                 critic_states=f(s[:, cidx]), critic_next_states=f(s2[:, cidx]))
    sel = lambda x, cols, n: ctx.select_columns(_t(x, dev), _t(cols, dev, np.int32), torch.empty(B, n, device=dev))
    sp, s2p, sc, s2c = sel(s, pidx, Op), sel(s2, pidx, Op), sel(s, cidx, Oc), sel(s2, cidx, Oc)
    pd = mlp_desc(ps.in_dim, ps.hidden, ps.out_dim, ps.act, ps.ln_first, False)
    qd = mlp_desc(qs.in_dim, qs.hidden, qs.out_dim, qs.act, qs.ln_first, False)
    P, Q, QT = _t(pp, dev), _t(qp, dev), _t(qtp, dev)
    LA = _t(np.array([log_alpha]), dev)
    pm, pv, qm, qv = (torch.zeros_like(x) for x in (P, P, Q, Q))
    am, av = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    hp = SacHparams(gamma, tau, -float(A), -20.0, 2.0, lr, lr, lr, 0.9, 0.999, 1e-8, sched)
    hp.critic_states, hp.critic_next_states = sc.data_ptr(), s2c.data_ptr()
    met = torch.zeros(10, device=dev)
    new_key, cnt = ctx.sac_update(pd, P, pm, pv, qd, Q, qm, qv, QT, LA, am, av, (sp, s2p, _t(a, dev), _t(r, dev), _t(term, dev)),
                                  key, 0, hp, met, 1)
    assert np.array_equal(new_key, new_key_e) and cnt == 1
    m = met.cpu().numpy()
    names = ["loss/q_loss", "loss/policy_loss", "loss/entropy_loss", "entropy/entropy", "entropy/alpha", "q_value/q_value"]
    for i, n in enumerate(names):
        assert m[i] == pytest.approx(float(met_e[n]), rel=1e-5, abs=1e-5), n
    assert np.linalg.norm(pm.cpu().numpy() * 10 - gp_e) / np.linalg.norm(gp_e) < 1e-5
    assert np.linalg.norm(qm.cpu().numpy() * 10 - gq_e) / np.linalg.norm(gq_e) < 1e-5


def _plugin(alg, env_over, alg_over, pidx, cidx):
    from rlx_amd.runner.config_dict import ConfigDict
    from rlx_amd.runner.default_config import get_config as runner_cfg
    import rlx_amd.algorithms.ppo.hip, rlx_amd.algorithms.sac.hip, rlx_amd.algorithms.ppo_lstm.hip  # noqa: F401,E401
    import rlx_amd.environments.synthetic.random_obs  # noqa: F401
    from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
    from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env
    config = ConfigDict()
    config.runner = runner_cfg("train")
    config.algorithm = get_algorithm_config(alg)
    config.environment = get_environment_config("synthetic.random_obs")
    for k, v in env_over.items():
        config.environment[k] = v
    for k, v in alg_over.items():
        config.algorithm[k] = v
    env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
    env.policy_observation_indices, env.critic_observation_indices = pidx, cidx      # what an asymmetric env exposes
    return get_algorithm_model_class(alg), config, env


def test_plugins_on_an_env_with_observation_index_sets(dev):
    pidx, cidx = np.arange(0, 12), np.arange(8, 40)
    # ---- PPO: two iterations; the nets are built on the selected widths and train
    cls, config, env = _plugin("ppo.hip", dict(nr_envs=64, obs_dim=40, act_dim=4, horizon=20),
                               dict(nr_steps=16, minibatch_size=256, nr_epochs=2, total_timesteps=2 * 64 * 16), pidx, cidx)
    m = cls(config, env, env, "/tmp/rlx_oi", None)
    assert (m.pdesc.in_dim, m.cdesc.in_dim) == (12, 32) and not m.use_fused_rollout
    p0, c0 = m.pparams.clone(), m.cparams.clone()
    m.train()
    assert all(np.isfinite(v) for v in m.last_metrics.values())
    assert (m.pparams - p0).abs().max().item() > 0 and (m.cparams - c0).abs().max().item() > 0
    assert len(m.evaluate(2)[0]) == 2
    # the rows the update saw are the selected columns of what the env produced
    batch = m._alloc_batch()
    state, _ = env.reset()
    full0 = state.clone()
    m.collect_rollout(batch, state.contiguous())
    assert torch.equal(batch.states[0], full0[:, torch.from_numpy(pidx).to(dev)])
    assert torch.equal(batch.cstates[0], full0[:, torch.from_numpy(cidx).to(dev)])
    # ---- SAC: a few vector steps + updates
    cls, config, env = _plugin("sac.hip", dict(nr_envs=32, obs_dim=40, act_dim=4),
                               dict(batch_size=64, buffer_size=32 * 64, learning_starts=64, total_timesteps=32 * 12,
                                    logging_frequency=32 * 4), pidx, cidx)
    s = cls(config, env, env, "/tmp/rlx_oi", None)
    assert (s.pdesc.in_dim, s.qdesc.in_dim) == (12, 32 + 4)
    s.train()
    assert all(np.isfinite(v) for v in s.last_metrics.values()) and s.opt_count > 0
    # ---- PPO+LSTM: the recurrent policy on its columns, the feed-forward critic on its own
    cls, config, env = _plugin("ppo_lstm.hip", dict(nr_envs=64, obs_dim=40, act_dim=4, horizon=20),
                               dict(nr_steps=16, minibatch_size=256, nr_epochs=2, total_timesteps=2 * 64 * 16,
                                    evaluation_and_save_frequency=-1), pidx, cidx)
    r = cls(config, env, env, "/tmp/rlx_oi", None)
    assert (r.ldesc.obs_dim, r.cdesc.in_dim) == (12, 32)
    p0, c0 = r.pparams.clone(), r.cparams.clone()
    r.train()
    assert all(np.isfinite(v) for v in r.last_metrics.values())
    assert (r.pparams - p0).abs().max().item() > 0 and (r.cparams - c0).abs().max().item() > 0
    batch = r._alloc_batch()
    state, _ = env.reset()
    full0 = state.clone()
    r.collect_rollout(batch, state.contiguous())
    assert torch.equal(batch.states[0], full0[:, torch.from_numpy(pidx).to(dev)])
    assert torch.equal(batch.cstates[0], full0[:, torch.from_numpy(cidx).to(dev)])
