"""GPU: PPO with a Categorical policy head (BASELINE.json configs[0], CartPole-shaped: the reference's only discrete PPO
head is DiscreteFlatValuesPolicy, rl_x/algorithms/ppo/pytorch/policy.py:96-135).  The HIP path against (i) the fixture
produced by executing that reference code in fp32 and (ii) the oracle on other shapes."""
import os

import numpy as np
import pytest
import torch

from oracle import discrete as odis, nets, ppo as oppo, prng
from rlx_amd.hip import PpoHparams, mlp_desc
from rlx_amd.hip import lib as L

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ACT_TANH = 0


def _t(a, dev, dtype=np.float32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(dev)


def _hp(clip, ec, cc, mgn):
    hp = PpoHparams(clip, ec, cc, mgn, 0.9, 0.999, 1e-8)
    hp.discrete_actions = 1
    return hp


@pytest.mark.parametrize("tag", ["f64r", "f32"])
def test_categorical_minibatch_updates_match_reference_fixture(ctx, dev, tag):
    """"f64r": the reference in float64 on float32-representable inputs -- the 1e-5 bar; "f32": the reference in fp32
    (its own rounding adds to ours: 2e-5)."""
    g = np.load(os.path.join(GOLDEN, "reference_ppo_discrete_%s.npz" % tag))
    lt, kt = (5e-6, 1e-5) if tag == "f64r" else (1e-5, 1e-4)   # gradients: 2 * lt
    O, NA, H = int(g["obs_dim"]), int(g["nr_actions"]), int(g["hidden"])
    pd = mlp_desc(O, [H, H], NA, ACT_TANH, False, False)
    cd = mlp_desc(O, [H, H], 1, ACT_TANH, False, False)
    B = g["states"].shape[0]
    states = _t(g["states"], dev)
    logits = torch.empty(B, NA, device=dev)
    ctx.mlp_fwd(pd, _t(g["pparams0"], dev), states, logits)
    np.testing.assert_allclose(logits.cpu().numpy(), g["logits"], rtol=1e-5, atol=2e-6)
    clip, ec, cc, mgn, lr = (float(g[k]) for k in ("clip_range", "entropy_coef", "critic_coef", "max_grad_norm", "learning_rate"))
    hp = _hp(clip, ec, cc, mgn)
    P, C = _t(g["pparams1"], dev), _t(g["cparams0"], dev)
    pm, pv, cm, cv = (torch.zeros_like(x) for x in (P, P, C, C))
    actions = _t(g["actions"].astype(np.float32).reshape(B, 1), dev)             # ONE float per sample: the action index
    logp, adv, ret = _t(g["log_probs"], dev), _t(g["advantages"], dev), _t(g["returns"], dev)
    for step in range(2):
        s = "_%d" % step
        idx = g["idx" + s]
        a = g["advantages"][idx].astype(np.float64)
        n = a.size
        stats = torch.tensor([a.sum(), n * (a.var(ddof=1) + a.mean() ** 2), float(n), 0.0], dtype=torch.float64, device=dev)
        pg, cg, met = torch.empty_like(P), torch.empty_like(C), torch.empty(8, device=dev)
        ctx.ppo_minibatch_fwd_bwd(pd, P, pg, cd, C, cg, met, states, actions, logp, ret, adv, _t(idx, dev, np.int32), hp,
                                  mb_global=n, stats_io=stats, phase=2)          # torch's unbiased std through the statistics
        m = met.cpu().numpy()
        np.testing.assert_allclose(m[0], g["pg_loss" + s], rtol=2 * lt, atol=2e-7)
        np.testing.assert_allclose(cc * m[1], g["critic_loss" + s], rtol=2 * lt, atol=2e-7)
        np.testing.assert_allclose(m[2], g["entropy_loss" + s], rtol=2 * lt, atol=2e-7)   # mean per-sample entropy
        np.testing.assert_allclose(m[3], g["approx_kl" + s], rtol=kt, atol=2e-8)
        np.testing.assert_allclose(m[4], g["clip_fraction" + s], atol=1.5 / n)
        for grads, name, norm in ((pg, "pgrads_clipped", float(g["policy_grad_norm" + s])), (cg, "cgrads_clipped", float(g["critic_grad_norm" + s]))):
            exp = g[name + s].astype(np.float64)
            got = grads.cpu().numpy().astype(np.float64) * min(1.0, mgn / (norm + 1e-6))
            assert np.linalg.norm(got - exp) / np.linalg.norm(exp) < 2 * lt
        ctx.clip_adam_step(P, pg, pm, pv, step + 1, lr, mgn)
        ctx.clip_adam_step(C, cg, cm, cv, step + 1, lr, mgn)
        d = np.abs(P.cpu().numpy() - g["pparams_after" + s])
        assert d.max() <= 2 * lr * (step + 1) and (d < 2e-6).mean() > 0.99


@pytest.mark.parametrize("O,NA,H,B,mb", [(4, 2, 64, 1024, 64), (9, 5, 128, 600, 200), (17, 8, 256, 4096, 2048)])
def test_categorical_minibatch_matches_oracle(ctx, dev, O, NA, H, B, mb):
    rng = np.random.default_rng(B + NA)
    ps, cs = nets.make_spec("A", O, NA, False, H), nets.make_spec("A", O, 1, False, H)
    pp = (nets.init_params(ps, rng, 0.01) + 0.1 * rng.standard_normal(ps.n_params)).astype(np.float32)
    cp = (nets.init_params(cs, rng, 1.0) + 0.05 * rng.standard_normal(cs.n_params)).astype(np.float32)
    states = rng.standard_normal((B, O)).astype(np.float32)
    lg, _ = nets.forward(ps, pp.astype(np.float64), states.astype(np.float64))
    actions = rng.integers(0, NA, size=B)
    lp, _ = odis.categorical_logp_entropy(lg, actions)
    logp = (lp + 0.1 * rng.standard_normal(B)).astype(np.float32)
    ret, adv = rng.standard_normal(B).astype(np.float32), (2 * rng.standard_normal(B) + 0.3).astype(np.float32)
    idx = rng.permutation(B)[:mb].astype(np.int32)
    f64 = lambda x: x.astype(np.float64)
    madv = oppo.normalize_advantages(f64(adv[idx]))
    clip, ec, cc = 0.2, 0.02, 0.5
    loss_e, met_e, gp_e, gc_e = odis.ppo_loss_and_grads(ps, f64(pp), cs, f64(cp), f64(states[idx]), actions[idx], f64(logp[idx]),
                                                        f64(ret[idx]), madv, clip, ec, cc)
    pd, cd = mlp_desc(O, [H, H], NA, ACT_TANH, False, False), mlp_desc(O, [H, H], 1, ACT_TANH, False, False)
    pg, cg, met = torch.zeros(ps.n_params, device=dev), torch.zeros(cs.n_params, device=dev), torch.zeros(8, device=dev)
    ctx.ppo_minibatch_fwd_bwd(pd, _t(pp, dev), pg, cd, _t(cp, dev), cg, met, _t(states, dev), _t(actions.reshape(B, 1), dev),
                              _t(logp, dev), _t(ret, dev), _t(adv, dev), _t(idx, dev, np.int32), _hp(clip, ec, cc, 0.5))
    m = met.cpu().numpy()
    np.testing.assert_allclose(m[0], met_e["loss/policy_gradient_loss"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(m[1], met_e["loss/critic_loss"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(m[2], met_e["loss/entropy_loss"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(m[3], met_e["policy_ratio/approx_kl"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(m[4], met_e["policy_ratio/clip_fraction"], atol=1.5 / mb)
    assert np.linalg.norm(pg.cpu().numpy() - gp_e) / np.linalg.norm(gp_e) < 2e-5
    assert np.linalg.norm(cg.cpu().numpy() - gc_e) / np.linalg.norm(gc_e) < 2e-5


@pytest.mark.parametrize("scheme", [1, 0])
def test_categorical_acting_matches_oracle(ctx, dev, scheme):
    O, NA, H, N = 4, 3, 64, 500
    rng = np.random.default_rng(2)
    ps, cs = nets.make_spec("A", O, NA, False, H), nets.make_spec("A", O, 1, False, H)
    pp = (nets.init_params(ps, rng, 0.01) + 0.3 * rng.standard_normal(ps.n_params)).astype(np.float32)
    cp = nets.init_params(cs, rng, 1.0).astype(np.float32)
    obs = rng.standard_normal((N, O)).astype(np.float32)
    pd, cd = mlp_desc(O, [H, H], NA, ACT_TANH, False, False), mlp_desc(O, [H, H], 1, ACT_TANH, False, False)
    key = prng.prng_key(5)
    action, value, logp = (torch.empty(N, device=dev) for _ in range(3))
    new_key = ctx.actor_critic_fwd_sample_discrete(pd, _t(pp, dev), cd, _t(cp, dev), _t(obs, dev), key, action, value, logp, scheme=scheme)
    ks = prng.split(key, 2, bool(scheme))
    assert np.array_equal(new_key, ks[0])
    logits, _ = nets.forward(ps, pp.astype(np.float64), obs.astype(np.float64))
    exp_a = odis.sample_categorical(ks[1], logits.astype(np.float32), bool(scheme))
    got_a = action.cpu().numpy().astype(np.int64)
    assert (got_a == exp_a).mean() > 0.995                      # identical noise; fp32 logits may flip a near-tie
    lp, _ = odis.categorical_logp_entropy(logits, got_a)
    np.testing.assert_allclose(logp.cpu().numpy(), lp, rtol=1e-5, atol=2e-6)
    v, _ = nets.forward(cs, cp.astype(np.float64), obs.astype(np.float64))
    np.testing.assert_allclose(value.cpu().numpy(), v.reshape(-1), rtol=1e-5, atol=2e-6)
    assert len(np.unique(got_a)) == NA
    det = torch.empty(N, device=dev)
    k2 = ctx.actor_critic_fwd_sample_discrete(pd, _t(pp, dev), cd, _t(cp, dev), _t(obs, dev), key, det, value, logp, scheme=scheme,
                                              deterministic=True)
    assert np.array_equal(k2, key)
    assert (det.cpu().numpy().astype(np.int64) == np.argmax(logits, axis=1)).mean() > 0.995


def test_cartpole_config0_trains(monkeypatch):
    """BASELINE.json configs[0]: PPO on CartPole-v1 with 8 envs through the reference-style entry point
    (Runner -> registry -> ppo.hip, DISCRETE action space -> Categorical head, host NUMPY env).  A random policy keeps the
    pole up for ~22 steps; after 60 iterations of 8 x 128 steps (61 k env steps) the build balances it for the full 500
    steps on seeds 1-3 -- the test asks for 200."""
    import sys
    from rlx_amd.runner.runner import Runner
    iters, N, T = 60, 8, 128
    monkeypatch.setattr(sys, "argv", [
        "experiment.py", "--algorithm.name=ppo.hip", "--environment.name=classic.cart_pole_v1", "--runner.mode=train",
        f"--environment.nr_envs={N}", f"--algorithm.nr_steps={T}", "--algorithm.minibatch_size=64", "--algorithm.nr_epochs=10",
        "--algorithm.network_architecture=flax", "--algorithm.nr_hidden_units=64", "--algorithm.learning_rate=3e-4",
        "--algorithm.anneal_learning_rate=false", "--algorithm.entropy_coef=0.0", "--algorithm.critic_coef=0.5",
        "--algorithm.clip_range=0.2", "--algorithm.max_grad_norm=0.5", "--algorithm.gae_lambda=0.95",
        f"--algorithm.total_timesteps={N * T * iters}"])
    model = Runner().run()
    m = model.last_metrics
    assert m["steps/nr_env_steps"] == N * T * iters and model.opt_count == iters * 10 * (N * T // 64)
    for k, v in m.items():
        assert np.isfinite(v), k
    assert m["policy/std_dev"] == 0.0                      # logged as 0 for a Categorical policy (ppo/pytorch/ppo.py:310)
    assert 0.0 < m["loss/entropy_loss"] <= np.log(2) + 1e-4
    assert m["rollout/episode_length"] > 200.0, m["rollout/episode_length"]
    returns, lengths = model.evaluate(5)                   # deterministic (argmax) episodes
    assert np.mean(lengths) > 200.0
