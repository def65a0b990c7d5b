"""CPU: the oracle against golden vectors produced by EXECUTING the reference's own code.

tests/golden/reference_{ppo,sac}_{f64,f64r,f32}.npz were written by tests/golden/make_reference_golden.py, which runs the
reference's PyTorch flavour (network modules loaded by file path; the GAE / loss / optimiser closures of `PPO.train` and
`SAC.train` compiled from the reference file where it lies) on seeded inputs -- see that script's docstring for what
can and cannot run in the authoring container.  These are the only fixtures whose expected values come from reference
code rather than from the restatement itself; they pin

    oracle.nets.forward / backward (Dense + tanh, Dense + relu)    ppo/pytorch/policy.py:44-50, critic.py:27-33
    oracle.ppo.gaussian_log_prob, processed_action, entropy        ppo/pytorch/policy.py:60-93
    oracle.ppo.gae                                                  ppo/pytorch/ppo.py:109-118
    oracle.ppo.ppo_loss_and_grads (ratio, clip, KL, clip fraction,  ppo/pytorch/ppo.py:121-166
        value loss, all gradients)
    oracle.ppo.clip_by_global_norm + adam_step (two steps)          torch.optim.Adam as constructed at ppo.py:82-84
    oracle.sac.policy_forward / tanh_gaussian / loss_and_grads      sac/pytorch/policy.py:45-73, sac.py:90-166
    Adam on the three SAC parameter groups                          sac.py:74-76

The PyTorch flavour differs from the JAX flavours (the parity target) in exactly these documented details, which the
tests below apply on the TEST side and nowhere else: `Tensor.std()` is the unbiased estimator (jnp.std: population);
`clip_grad_norm_` scales by c / (norm + 1e-6) (optax: c / norm); `critic_loss` is reported times critic_coef."""
import os

import numpy as np
import pytest

from oracle import nets, ppo as oppo, prng, sac as osac

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _tol(tag):
    if tag == "f64r":       # "f64r": float64 arithmetic on float32-representable inputs; the stored intermediate results that
        return dict(rtol=2e-7, atol=1e-9)   # feed later stages (values, log-probs, advantages ...) are themselves rounded to fp32
    return dict(rtol=1e-10, atol=1e-12) if tag in ("f64", "f64r") else dict(rtol=2e-5, atol=2e-6)


def torch_flavour_normalize(a):
    """ppo/pytorch/ppo.py:135: (a - a.mean()) / (a.std() + 1e-8) with torch's UNBIASED std."""
    return (a - a.mean()) / (a.std(ddof=1) + 1e-8)


@pytest.mark.parametrize("tag", ["f64", "f64r", "f32"])
def test_ppo_networks_logprob_gae_match_reference(tag):
    g = _load("reference_ppo_%s.npz" % tag)
    assert str(g["source"]).startswith("reference:")
    O, A, H = int(g["obs_dim"]), int(g["act_dim"]), int(g["hidden"])
    ps, cs = nets.make_spec("A", O, A, True, H), nets.make_spec("A", O, 1, False, H)
    f = lambda k: g[k].astype(np.float64)
    pp, cp = f("pparams0"), f("cparams0")
    assert pp.size == ps.n_params and cp.size == cs.n_params
    T, N = g["rewards"].shape
    tol = _tol(tag)
    mean, _ = nets.forward(ps, pp, f("states").reshape(-1, O))
    np.testing.assert_allclose(mean.reshape(T, N, A), g["mean"], **tol)
    logstd = pp[ps.logstd:][None, :]
    np.testing.assert_allclose(oppo.gaussian_log_prob(f("actions").reshape(-1, A), mean, logstd).reshape(T, N), g["log_probs"], **tol)
    np.testing.assert_allclose(oppo.processed_action(f("actions"), True, -1.0, 1.0), g["scaled_actions"], **tol)
    np.testing.assert_allclose(oppo.processed_action(mean.reshape(T, N, A), True, -1.0, 1.0), g["deterministic_actions"], **tol)
    v, _ = nets.forward(cs, cp, f("states").reshape(-1, O))
    nv, _ = nets.forward(cs, cp, f("next_states").reshape(-1, O))
    np.testing.assert_allclose(v.reshape(T, N), g["values"], **tol)
    np.testing.assert_allclose(nv.reshape(T, N), g["next_values"], **tol)
    adv, ret = oppo.gae(f("rewards"), f("values"), f("next_values"), f("terminations"), float(g["gamma"]), float(g["gae_lambda"]))
    np.testing.assert_allclose(adv, g["advantages"], **tol)
    np.testing.assert_allclose(ret, g["returns"], **tol)


@pytest.mark.parametrize("tag", ["f64", "f64r", "f32"])
def test_ppo_minibatch_updates_match_reference(tag):
    g = _load("reference_ppo_%s.npz" % tag)
    O, A, H = int(g["obs_dim"]), int(g["act_dim"]), int(g["hidden"])
    ps, cs = nets.make_spec("A", O, A, True, H), nets.make_spec("A", O, 1, False, H)
    f = lambda k: g[k].astype(np.float64)
    clip, ec, cc, mgn, lr = (float(g[k]) for k in ("clip_range", "entropy_coef", "critic_coef", "max_grad_norm", "learning_rate"))
    pst, cst = oppo.TrainState(ps, f("pparams1")), oppo.TrainState(cs, f("cparams0"))
    bs, ba = f("states").reshape(-1, O), f("actions").reshape(-1, A)
    badv, bret, blp = f("advantages").reshape(-1), f("returns").reshape(-1), f("log_probs").reshape(-1)
    tol = _tol(tag)
    gtol = 1e-9 if tag in ("f64", "f64r") else 2e-5
    for step in range(2):
        s = "_%d" % step
        idx = g["idx" + s]
        madv = torch_flavour_normalize(badv[idx])
        _, m, gp, gc = oppo.ppo_loss_and_grads(ps, pst.params, cs, cst.params, bs[idx], ba[idx], blp[idx], bret[idx], madv, clip, ec, cc)
        np.testing.assert_allclose(m["loss/policy_gradient_loss"], g["pg_loss" + s], **tol)
        np.testing.assert_allclose(m["loss/entropy_loss"], g["entropy_loss" + s], **tol)
        np.testing.assert_allclose(m["policy_ratio/approx_kl"], g["approx_kl" + s], **tol)
        assert m["policy_ratio/clip_fraction"] == pytest.approx(float(g["clip_fraction" + s]), abs=1e-7)
        assert 0.0 < float(g["clip_fraction" + s]) < 1.0                       # the fixture does exercise the clip
        np.testing.assert_allclose(cc * m["loss/critic_loss"], g["critic_loss" + s], **tol)
        np.testing.assert_allclose(oppo.global_norm(gp), g["policy_grad_norm" + s], rtol=gtol)
        np.testing.assert_allclose(oppo.global_norm(gc), g["critic_grad_norm" + s], rtol=gtol)
        # gradients: the reference left the CLIPPED gradients in .grad (clip_grad_norm_ is in place)
        for got, name, norm in ((gp, "pgrads_clipped", float(g["policy_grad_norm" + s])), (gc, "cgrads_clipped", float(g["critic_grad_norm" + s]))):
            exp = g[name + s].astype(np.float64)
            scale = min(1.0, mgn / (norm + 1e-6))                              # torch's clip coefficient
            assert np.linalg.norm(got * scale - exp) / np.linalg.norm(exp) < gtol
        pst.apply_gradients(gp, lr, mgn)
        cst.apply_gradients(gc, lr, mgn)
        for st, name in ((pst, "pparams_after"), (cst, "cparams_after")):
            exp = g[name + s].astype(np.float64)
            d = np.abs(st.params - exp)
            if tag in ("f64", "f64r"):
                assert d.max() < 1e-9                                          # optax clip (c/norm) vs torch (c/(norm+1e-6)): <1e-6 relative on g
            else:
                assert d.max() <= 2 * lr * (step + 1) and (d < 2e-6).mean() > 0.99
    assert float(g["policy_grad_norm_0"]) > mgn                                # ... and the gradient clip was active


@pytest.mark.parametrize("tag", ["f64", "f64r", "f32"])
def test_sac_losses_gradients_and_adam_match_reference(tag):
    g = _load("reference_sac_%s.npz" % tag)
    assert str(g["source"]).startswith("reference:")
    O, A, H = int(g["obs_dim"]), int(g["act_dim"]), int(g["hidden"])
    ps, qs = osac.make_specs(O, A, H)
    f = lambda k: g[k].astype(np.float64)
    pp, qp, qtp = f("pparams"), f("qparams"), f("qtarget")
    assert pp.size == ps.n_params and qp.size == 2 * qs.n_params
    lo, hi = float(g["log_std_min"]), float(g["log_std_max"])
    tol = _tol(tag)
    gtol = 1e-9 if tag in ("f64", "f64r") else 5e-5
    nm, nls, _, _ = osac.policy_forward(ps, pp, f("next_states"), lo, hi)
    na, nlp = osac.tanh_gaussian(nm, nls, f("noise_next"))
    np.testing.assert_allclose(na, g["next_action"], **tol)
    np.testing.assert_allclose(nlp, g["next_log_prob"], rtol=tol["rtol"] * 10, atol=tol["atol"] * 10)
    np.testing.assert_allclose(np.tanh(nm), g["deterministic_action"], **tol)
    for k, name in enumerate(("q1", "q2")):
        np.testing.assert_allclose(osac.q_forward(qs, qp, k, f("states"), f("actions"))[0], g[name], **tol)
    met, gpol, gq, ga = osac.loss_and_grads(ps, pp, qs, qp, qtp, np.float64(g["log_alpha"]), f("states"), f("next_states"),
                                            f("actions"), f("rewards"), f("terminations"), f("noise_next"), f("noise_cur"),
                                            float(g["gamma"]), float(g["target_entropy"]), lo, hi)
    ltol = dict(rtol=1e-9, atol=1e-11) if tag in ("f64", "f64r") else dict(rtol=5e-5, atol=5e-6)
    np.testing.assert_allclose(met["loss/q_loss"], g["q_loss"], **ltol)
    np.testing.assert_allclose(met["loss/policy_loss"], g["policy_loss"], **ltol)
    np.testing.assert_allclose(met["loss/entropy_loss"], g["entropy_loss"], **ltol)
    np.testing.assert_allclose(met["entropy/entropy"], g["entropy"], **ltol)
    np.testing.assert_allclose(met["entropy/alpha"], g["alpha"], **ltol)
    np.testing.assert_allclose(met["q_value/q_value"], g["min_q_mean"], **ltol)
    assert np.linalg.norm(gq - f("gcritic")) / np.linalg.norm(f("gcritic")) < gtol
    assert np.linalg.norm(gpol - f("gpolicy")) / np.linalg.norm(f("gpolicy")) < gtol
    assert float(ga) == pytest.approx(float(g["g_log_alpha"]), rel=gtol * 10)
    n = qs.n_params
    assert np.linalg.norm(gq[:n]) + np.linalg.norm(gq[n:]) == pytest.approx(float(g["critic_grad_norm"]), rel=gtol * 10)
    assert np.linalg.norm(gpol) == pytest.approx(float(g["policy_grad_norm"]), rel=gtol * 10)
    # Adam (no gradient clip in SAC): first step of each of the three optimisers
    lr = float(g["learning_rate"])
    z = lambda x: (np.zeros_like(x), np.zeros_like(x))
    for p0, grad, name in ((pp, gpol, "pparams_after"), (qp, gq, "qparams_after")):
        p1, _, _ = oppo.adam_step(p0, grad, *z(p0), 0, lr)
        d = np.abs(p1 - f(name))
        if tag in ("f64", "f64r"):
            assert d.max() < 1e-9
        else:
            assert d.max() <= 2 * lr and (d < 2e-6).mean() > 0.99
    la1, _, _ = oppo.adam_step(np.array([float(g["log_alpha"])]), np.array([float(ga)]), np.zeros(1), np.zeros(1), 0, lr)
    assert la1[0] == pytest.approx(float(g["log_alpha_after"]), abs=1e-7)


def test_replay_ring_matches_reference():
    """oracle.sac.ReplayBuffer vs the JAX flavour's own numpy class (sac/flax/replay_buffer.py, executed): ring wrap,
    capacity rounding, PCG64 index draws, gathered rows -- bit for bit."""
    g = _load("reference_sac_replay.npz")
    assert str(g["source"]).startswith("reference:")
    NE, B = int(g["nr_envs"]), int(g["batch"])
    steps, _, O = g["add_states"].shape
    A = g["add_actions"].shape[2]
    rb = osac.ReplayBuffer(int(g["capacity"]), NE, O, A, np.random.default_rng(int(g["sampler_seed"])))
    names = ("states", "next_states", "actions", "rewards", "terminations")
    for t in range(steps):
        rb.add(*(g["add_" + k][t] for k in names))
        if ("sample%d_states" % t) in g.files:
            got = rb.gather(*rb.sample_indices(B))
            for k, x in zip(names, got):
                assert np.array_equal(x, g["sample%d_%s" % (t, k)]), (t, k)
    assert rb.pos == int(g["final_pos"]) and rb.size == int(g["final_size"])
    assert np.array_equal(rb.states, g["ring_states"])


@pytest.mark.parametrize("tag", ["f64", "f64r", "f32"])
def test_ppo_categorical_policy_matches_reference(tag):
    """oracle.discrete against the reference's DiscreteFlatValuesPolicy + policy_loss_fn / critic_loss_fn closures
    (ppo/pytorch/policy.py:96-135, ppo.py:121-166, executed): logits, log-probs, entropies, argmax actions, loss terms,
    all gradients, clip + Adam over two minibatches -- BASELINE.json configs[0]'s head."""
    from oracle import discrete as odis
    g = _load("reference_ppo_discrete_%s.npz" % tag)
    assert str(g["source"]).startswith("reference:")
    O, NA, H = int(g["obs_dim"]), int(g["nr_actions"]), int(g["hidden"])
    ps, cs = nets.make_spec("A", O, NA, False, H), nets.make_spec("A", O, 1, False, H)
    f = lambda k: g[k].astype(np.float64)
    tol = _tol(tag)
    gtol = 1e-9 if tag in ("f64", "f64r") else 2e-5
    logits, _ = nets.forward(ps, f("pparams0"), f("states"))
    np.testing.assert_allclose(logits, g["logits"], **tol)
    lp, _ = odis.categorical_logp_entropy(logits, g["actions"])
    np.testing.assert_allclose(lp, g["log_probs"], **tol)
    assert np.array_equal(np.argmax(logits, axis=1), g["deterministic_actions"])
    clip, ec, cc, mgn, lr = (float(g[k]) for k in ("clip_range", "entropy_coef", "critic_coef", "max_grad_norm", "learning_rate"))
    pst, cst = oppo.TrainState(ps, f("pparams1")), oppo.TrainState(cs, f("cparams0"))
    for step in range(2):
        s = "_%d" % step
        idx = g["idx" + s]
        lg, _ = nets.forward(ps, pst.params, f("states")[idx])
        nlp, ent = odis.categorical_logp_entropy(lg, g["actions"][idx])
        np.testing.assert_allclose(nlp, g["new_log_prob" + s], **tol)
        np.testing.assert_allclose(ent, g["entropy" + s], **tol)
        madv = torch_flavour_normalize(f("advantages")[idx])
        _, m, gp, gc = odis.ppo_loss_and_grads(ps, pst.params, cs, cst.params, f("states")[idx], g["actions"][idx],
                                               f("log_probs")[idx], f("returns")[idx], madv, clip, ec, cc)
        np.testing.assert_allclose(m["loss/policy_gradient_loss"], g["pg_loss" + s], **tol)
        np.testing.assert_allclose(m["loss/entropy_loss"], g["entropy_loss" + s], **tol)
        np.testing.assert_allclose(m["policy_ratio/approx_kl"], g["approx_kl" + s], **tol)
        assert m["policy_ratio/clip_fraction"] == pytest.approx(float(g["clip_fraction" + s]), abs=1e-7)
        np.testing.assert_allclose(cc * m["loss/critic_loss"], g["critic_loss" + s], **tol)
        for got, name, norm in ((gp, "pgrads_clipped", float(g["policy_grad_norm" + s])), (gc, "cgrads_clipped", float(g["critic_grad_norm" + s]))):
            np.testing.assert_allclose(oppo.global_norm(got), norm, rtol=gtol)
            exp = g[name + s].astype(np.float64)
            assert np.linalg.norm(got * min(1.0, mgn / (norm + 1e-6)) - exp) / np.linalg.norm(exp) < gtol
        pst.apply_gradients(gp, lr, mgn)
        cst.apply_gradients(gc, lr, mgn)
        d = np.abs(pst.params - g["pparams_after" + s].astype(np.float64))
        assert d.max() < (1e-9 if tag in ("f64", "f64r") else 2 * lr * (step + 1))
    assert 0.0 < float(g["clip_fraction_0"]) < 1.0 and float(g["policy_grad_norm_0"]) > mgn


def test_sample_categorical_is_gumbel_argmax():
    """jax.random.categorical restated: frequencies follow softmax(logits); same key -> same draw."""
    from oracle import discrete as odis
    logits = np.tile(np.array([[0.0, 1.0, -1.0, 0.5]], np.float32), (20000, 1))
    a = odis.sample_categorical(prng.prng_key(3), logits)
    assert np.array_equal(a, odis.sample_categorical(prng.prng_key(3), logits))
    freq = np.bincount(a, minlength=4) / len(a)
    p = np.exp(logits[0]) / np.exp(logits[0]).sum()
    assert np.abs(freq - p).max() < 0.012
