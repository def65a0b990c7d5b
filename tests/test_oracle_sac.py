"""CPU: the SAC oracle's manual backward vs float64 torch.autograd; replay-buffer semantics."""
import numpy as np
import pytest

from oracle import prng, sac


def _problem(O, A, B, H, seed=0):
    rng = np.random.default_rng(seed)
    ps, qs = sac.make_specs(O, A, H)
    pp = sac.lecun_normal_init(ps, rng, np.float64) + 0.02 * rng.standard_normal(ps.n_params)
    qp = np.concatenate([sac.lecun_normal_init(qs, rng, np.float64) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)
    qtp = qp + 0.01 * rng.standard_normal(qp.shape)
    s, s2 = rng.standard_normal((B, O)), rng.standard_normal((B, O))
    a = np.tanh(rng.standard_normal((B, A)))
    r = rng.standard_normal(B)
    term = (rng.random(B) < 0.2).astype(np.float64)
    return ps, qs, pp, qp, qtp, s, s2, a, r, term


def test_manual_backward_matches_autograd():
    O, A, B = 11, 3, 40
    ps, qs, pp, qp, qtp, s, s2, a, r, term = _problem(O, A, B, 64)
    key, e1, e2 = sac.sample_noise(prng.prng_key(3), B, A)
    met, gp, gc, ga = sac.loss_and_grads(ps, pp, qs, qp, qtp, np.float64(-0.3), s, s2, a, r, term, e1.astype(np.float64),
                                         e2.astype(np.float64), 0.99, -A)
    l2, gp2, gc2, ga2 = sac.loss_torch(ps, pp, qs, qp, qtp, -0.3, s, s2, a, r, term, e1, e2, 0.99, -A)
    assert abs(met["loss/q_loss"] + met["loss/policy_loss"] + met["loss/entropy_loss"] - l2) < 1e-12
    np.testing.assert_allclose(gp, gp2, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(gc, gc2, rtol=1e-9, atol=1e-13)
    assert abs(ga - ga2) < 1e-12


def test_per_sample_noise_keys():
    # keys = split(key, 2B+1): key <- keys[0], sample i uses keys[1+2i] / keys[2+2i]  (sac.py:196-197)
    key = prng.prng_key(7)
    new_key, e1, e2 = sac.sample_noise(key, 4, 3)
    keys = prng.split(key, 9)
    assert np.array_equal(new_key, keys[0])
    assert np.array_equal(e1[2], prng.normal(keys[5], (3,)))
    assert np.array_equal(e2[2], prng.normal(keys[6], (3,)))


def test_replay_buffer_semantics():
    rb = sac.ReplayBuffer(40, 4, 3, 2, np.random.default_rng(0))
    assert rb.capacity == 10
    for t in range(13):
        rb.add(np.full((4, 3), t), np.full((4, 3), t + 1), np.full((4, 2), t), np.full(4, t), np.zeros(4))
    assert rb.size == 10 and rb.pos == 3
    assert rb.states[0, 0, 0] == 10 and rb.states[3, 0, 0] == 3     # ring overwrote rows 0..2
    i1, i2 = rb.sample_indices(64)
    assert i1.max() < 10 and i2.max() < 4
    s, s2, a, r, tm = rb.gather(i1, i2)
    np.testing.assert_array_equal(s2[:, 0], s[:, 0] + 1)


def test_golden_sac_matches_oracle():
    """tests/golden/sac.npz (make_golden.py, source "restatement") against a fresh evaluation of the oracle."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sac.npz"))
    O, A, H = int(g["obs_dim"]), int(g["act_dim"]), int(g["hidden"])
    ps, qs = sac.make_specs(O, A, H)
    f = lambda x: np.asarray(x, dtype=np.float64)
    nk, e1, e2 = sac.sample_noise(g["key"], g["states"].shape[0], A, True)
    assert np.array_equal(nk, g["new_key"])
    np.testing.assert_array_equal(e1.astype(np.float32), g["eps_next"])
    np.testing.assert_array_equal(e2.astype(np.float32), g["eps_cur"])
    met, gp, gq, ga = sac.loss_and_grads(ps, f(g["pparams"]), qs, f(g["qparams"]), f(g["qtarget"]), np.float64(g["log_alpha"]),
                                         f(g["states"]), f(g["next_states"]), f(g["actions"]), f(g["rewards"]),
                                         f(g["terminations"]), f(e1), f(e2), float(g["gamma"]), float(g["target_entropy"]))
    assert met["loss/q_loss"] == pytest.approx(float(g["q_loss"]), rel=1e-12)
    assert met["loss/policy_loss"] == pytest.approx(float(g["policy_loss"]), rel=1e-12)
    np.testing.assert_allclose(gp, g["gpolicy"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(gq, g["gcritic"], rtol=1e-6, atol=1e-9)


def test_full_jit_nets_manual_backward_matches_autograd():
    """The 512-LayerNorm-256-128 ELU nets of sac/flax_full_jit through the same loss: manual reverse pass == float64 autograd."""
    rng = np.random.default_rng(3)
    O, A, B = 9, 3, 24
    ps, qs = sac.make_specs(O, A, arch="full_jit")
    assert ps.hidden == [512, 256, 128] and ps.ln_first and qs.in_dim == O + A
    f = np.float64
    pp = sac.lecun_normal_init(ps, rng, f) + 0.02 * rng.standard_normal(ps.n_params)
    pp[ps.head["W"]:ps.head["W"] + ps.head["in"] * ps.head["out"]] *= 0.1
    qp = np.concatenate([sac.lecun_normal_init(qs, rng, f) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)
    qtp = qp + 0.01 * rng.standard_normal(qp.shape)
    s, s2 = rng.standard_normal((B, O)), rng.standard_normal((B, O))
    a = np.tanh(rng.standard_normal((B, A)))
    r, term = rng.standard_normal(B), (rng.random(B) < 0.2).astype(f)
    key = prng.prng_key(5)
    _, e1, e2 = sac.sample_noise(key, B, A, True, schedule=1)
    met, gp, gq, ga = sac.loss_and_grads(ps, pp, qs, qp, qtp, f(-0.3), s, s2, a, r, term, e1.astype(f), e2.astype(f), 0.99, -float(A))
    _, gp_t, gq_t, ga_t = sac.loss_torch(ps, pp, qs, qp, qtp, f(-0.3), s, s2, a, r, term, e1.astype(f), e2.astype(f), 0.99, -float(A))
    assert np.abs(gp - gp_t).max() <= 1e-11 * max(1.0, np.abs(gp_t).max())
    assert np.abs(gq - gq_t).max() <= 1e-11 * max(1.0, np.abs(gq_t).max())
    assert abs(ga - ga_t) <= 1e-12


def test_full_jit_key_schedule_and_replay_indices():
    """keys = split(key, 2B+2): [0] next key, [1] replay key, then two contiguous blocks of per-sample noise keys; both
    replay index vectors are drawn from the SAME key (sac/flax_full_jit/sac.py:273-282)."""
    key = prng.prng_key(7)
    B, A = 6, 2
    keys = prng.split(key, 2 * B + 2)
    nk, e1, e2 = sac.sample_noise(key, B, A, True, schedule=1)
    assert np.array_equal(nk, keys[0])
    assert np.array_equal(e1[3], prng.normal(keys[2 + 3], (A,))) and np.array_equal(e2[0], prng.normal(keys[2 + B], (A,)))
    nk0, f1, f2 = sac.sample_noise(key, B, A, True, schedule=0)
    assert np.array_equal(nk0, prng.split(key, 2 * B + 1)[0]) and not np.array_equal(e1, f1)
    i1, i2 = sac.replay_indices(key, B, 40, 5)
    assert np.array_equal(i1, prng.randint(keys[1], (B,), 0, 40)) and np.array_equal(i2, prng.randint(keys[1], (B,), 0, 5))
    assert i1.min() >= 0 and i1.max() < 40 and i2.max() < 5
    big1, _ = sac.replay_indices(key, 4096, 244, 4096)
    assert abs(big1.mean() - 121.5) < 5 and big1.max() == 243
