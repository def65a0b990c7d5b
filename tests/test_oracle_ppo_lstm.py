"""CPU: the recurrent-policy oracle (oracle/ppo_lstm.py).  The reference has no tests for this path and JAX is not
installable here, so these pin the oracle through the properties the reference's code relies on."""
import numpy as np
import torch

from oracle import ppo_lstm as ol, prng


def _spec_params(rng, share=False):
    spec = ol.LstmPolicySpec(5, 3, 64, 64, (64, 32, 16), share)
    p = ol.init_params(spec, rng) + 0.05 * rng.standard_normal(spec.n_params)
    return spec, torch.tensor(p)


def test_sequence_replay_reproduces_the_rollout():
    """ppo_lstm.py:148-149 masks the carry AFTER the env step; policy.py:134-142 masks with done[t-1] BEFORE obs[t]:
    replaying the stored observations/dones from the stored initial carry must give the rollout's means."""
    rng = np.random.default_rng(0)
    for share in (False, True):
        spec, p = _spec_params(rng, share)
        T, n = 7, 6
        obs = torch.tensor(rng.standard_normal((T, n, spec.O)))
        done = torch.tensor((rng.random((T, n)) < 0.3).astype(np.float64))
        c = torch.tensor(rng.standard_normal((n, spec.H)))
        h = torch.tanh(torch.tensor(rng.standard_normal((n, spec.H))))
        c0, h0 = c.clone(), h.clone()
        means = []
        for t in range(T):
            mean, c, h = ol.apply_one_step(spec, p, obs[t], c, h)
            m = (1.0 - done[t])[:, None]
            c, h = c * m, h * m
            means.append(mean)
        seq = ol.forward_sequence(spec, p, obs, done, c0, h0)
        assert torch.allclose(torch.stack(means), seq, atol=1e-12)


def test_lstm_cell_matches_torch_lstmcell():
    """OptimizedLSTMCell's gate order (i, f, g, o) and carry order (c, h) == torch.nn.LSTMCell with the weights mapped."""
    rng = np.random.default_rng(1)
    spec, p = _spec_params(rng)
    E, H = spec.E, spec.H
    cell = torch.nn.LSTMCell(E, H).double()
    with torch.no_grad():
        cell.weight_ih.copy_(spec.get(p, "lstm.Wi", (E, 4 * H)).T)
        cell.weight_hh.copy_(spec.get(p, "lstm.Wh", (H, 4 * H)).T)
        cell.bias_ih.zero_()
        cell.bias_hh.copy_(spec.get(p, "lstm.bh"))
    x = torch.tensor(rng.standard_normal((9, E)))
    c = torch.tensor(rng.standard_normal((9, H)))
    h = torch.tensor(rng.standard_normal((9, H)))
    c2, h2 = ol.lstm_cell(spec, p, c, h, x)
    h_ref, c_ref = cell(x, (h, c))
    assert torch.allclose(c2, c_ref, atol=1e-12) and torch.allclose(h2, h_ref, atol=1e-12)


def test_gru_cell_matches_torch_grucell_and_sequence_replay():
    """flax.linen.GRUCell (r, z, n gate order; bias on the input projections and on hn only) == torch.nn.GRUCell with
    the weights mapped; and the sequence replay reproduces the step-by-step rollout with carry resets."""
    rng = np.random.default_rng(4)
    spec = ol.LstmPolicySpec(5, 3, 64, 64, (64, 32, 16), False, "gru")
    p = torch.tensor(ol.init_params(spec, rng) + 0.05 * rng.standard_normal(spec.n_params))
    E, H = spec.E, spec.H
    cell = torch.nn.GRUCell(E, H).double()
    with torch.no_grad():
        cell.weight_ih.copy_(spec.get(p, "gru.Wi", (E, 3 * H)).T)
        cell.bias_ih.copy_(spec.get(p, "gru.bi"))
        cell.weight_hh.copy_(torch.cat([spec.get(p, "gru.Wh_rz", (H, 2 * H)), spec.get(p, "gru.Wh_n", (H, H))], dim=1).T)
        b = torch.zeros(3 * H, dtype=torch.float64)
        b[2 * H:] = spec.get(p, "gru.bhn")
        cell.bias_hh.copy_(b)
    x, h = torch.tensor(rng.standard_normal((9, E))), torch.tensor(rng.standard_normal((9, H)))
    assert torch.allclose(ol.gru_cell(spec, p, h, x), cell(x, h), atol=1e-12)
    T, n = 6, 5
    obs = torch.tensor(rng.standard_normal((T, n, spec.O)))
    done = torch.tensor((rng.random((T, n)) < 0.3).astype(np.float64))
    hh = torch.tanh(torch.tensor(rng.standard_normal((n, H))))
    c = torch.zeros(n, H, dtype=torch.float64)
    h0, means = hh.clone(), []
    for t in range(T):
        mean, c, hh = ol.apply_one_step(spec, p, obs[t], c, hh)
        hh = hh * (1.0 - done[t])[:, None]
        means.append(mean)
    assert torch.allclose(torch.stack(means), ol.forward_sequence(spec, p, obs, done, torch.zeros_like(h0), h0), atol=1e-12)


def test_env_minibatch_indices_are_row_permutations():
    for part in (True, False):
        key, idx = ol.env_minibatch_indices(prng.prng_key(3), 48, 5, 4, 12, part)
        assert idx.shape == (20, 12)
        rows = idx.reshape(5, 48)
        for r in rows:
            assert np.array_equal(np.sort(r), np.arange(48))
        assert not np.array_equal(rows[0], rows[1])
        assert np.array_equal(key, prng.split(prng.prng_key(3), 2, part)[0])


def test_loss_is_mean_over_time_and_envs():
    rng = np.random.default_rng(2)
    from oracle import nets
    spec, p = _spec_params(rng)
    cs = nets.MLPSpec(spec.O, [16, 8], 1, nets.ACT_ELU, True, False)
    cp = torch.tensor(nets.init_params(cs, rng, 1.0, dtype=np.float64))
    T, n = 4, 5
    t = lambda a: torch.tensor(a)
    obs, act = t(rng.standard_normal((T, n, spec.O))), t(rng.standard_normal((T, n, spec.A)))
    logp, ret, adv = (t(rng.standard_normal((T, n))) for _ in range(3))
    done = t((rng.random((T, n)) < 0.2).astype(np.float64))
    c0, h0 = t(rng.standard_normal((n, spec.H))), t(rng.standard_normal((n, spec.H)))
    loss, met = ol.ppo_lstm_loss(spec, p, cs, cp, obs, act, logp, ret, adv, done, c0, h0, 0.2, 0.01, 0.5)
    # per-env losses (the reference vmaps loss_fn over envs, then means): same number
    per_env = [ol.ppo_lstm_loss(spec, p, cs, cp, obs[:, e:e + 1], act[:, e:e + 1], logp[:, e:e + 1], ret[:, e:e + 1],
                                adv[:, e:e + 1], done[:, e:e + 1], c0[e:e + 1], h0[e:e + 1], 0.2, 0.01, 0.5)[0] for e in range(n)]
    assert torch.allclose(loss, torch.stack(per_env).mean(), atol=1e-12)
    expected = met["loss/policy_gradient_loss"] - 0.01 * met["loss/entropy_loss"] + 0.5 * met["loss/critic_loss"]
    assert torch.allclose(loss, expected, atol=1e-12)


def test_golden_ppo_lstm_matches_oracle():
    """tests/golden/ppo_lstm.npz against a fresh evaluation of the oracle (loss terms, BPTT gradients, env-index permutation)."""
    import os
    from oracle import nets
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ppo_lstm.npz"))
    O, A = int(g["obs_dim"]), int(g["act_dim"])
    spec = ol.LstmPolicySpec(O, A, 128, 64, (512, 256, 128), False)
    cs = nets.make_spec("B", O, 1, False)
    t64 = lambda x: torch.tensor(np.asarray(x, dtype=np.float64))
    ei = g["env_idx"]
    a = g["advantages"][:, ei].astype(np.float64)
    a = (a - a.mean()) / (a.std() + 1e-8)
    P, C = t64(g["pparams"]).requires_grad_(True), t64(g["cparams"]).requires_grad_(True)
    loss, met = ol.ppo_lstm_loss(spec, P, cs, C, t64(g["states"][:, ei]), t64(g["actions"][:, ei]), t64(g["log_probs"][:, ei]),
                                 t64(g["returns"][:, ei]), t64(a), t64(g["dones"][:, ei]), t64(g["c0"][ei]), t64(g["h0"][ei]),
                                 float(g["clip_range"]), float(g["entropy_coef"]), float(g["critic_coef"]))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-12
    assert abs(met["policy_ratio/approx_kl"].item() - float(g["approx_kl"])) < 1e-12
    np.testing.assert_allclose(P.grad.numpy(), g["pgrads"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(C.grad.numpy(), g["cgrads"], rtol=1e-6, atol=1e-9)
    nk, idx = ol.env_minibatch_indices(g["key"], g["states"].shape[1], 2, g["states"].shape[1] // 8, 8)
    assert np.array_equal(nk, g["perm_key"]) and np.array_equal(idx, g["perm_env_idx"])


def test_film_combine_is_an_affine_modulation_of_the_obs_latent():
    """lstm_obs_combine_method = "film" (ppo_lstm/flax_full_jit/policy.py:55-57,95-100): gamma = Dense(E)(latent),
    beta = Dense(E)(latent), torso input = obs_latent * gamma + beta.  With a zero FiLM kernel and bias (1 | 0) the
    modulation is the identity: the decoder then sees the obs latent alone (E wide), whatever the carry holds."""
    import torch
    rng = np.random.default_rng(0)
    spec = ol.LstmPolicySpec(5, 2, 64, 64, (64, 32, 16), False, "lstm", "film")
    assert spec.K1 == 64 and spec.off["t1.W"][1] == 64 * 64 and spec.off["film.W"][1] == 64 * 128
    p = torch.tensor(ol.init_params(spec, rng, 1.0) + 0.05 * rng.standard_normal(spec.n_params))
    o, n = spec.off["film.W"]
    p[o:o + n] = 0.0
    ob, nb = spec.off["film.b"]
    p[ob:ob + 64] = 1.0
    p[ob + 64:ob + nb] = 0.0
    obs = torch.tensor(rng.standard_normal((7, 5)))
    lat_o = ol._encode(spec, p, obs, "enc_o")
    m1 = ol.decode(spec, p, lat_o, torch.tensor(rng.standard_normal((7, 64))))
    m2 = ol.decode(spec, p, lat_o, torch.tensor(rng.standard_normal((7, 64))))
    assert torch.allclose(m1, m2, atol=1e-14)
    # a generic FiLM layer: check against the formula written out
    p2 = torch.tensor(ol.init_params(spec, rng, 1.0) + 0.05 * rng.standard_normal(spec.n_params))
    h = torch.tensor(rng.standard_normal((7, 64)))
    lat = torch.nn.functional.elu(ol._ln(h, spec.get(p2, "lstm_ln.g"), spec.get(p2, "lstm_ln.be")))
    W, b = spec.get(p2, "film.W", (64, 128)), spec.get(p2, "film.b")
    x = lat_o * (lat @ W[:, :64] + b[:64]) + (lat @ W[:, 64:] + b[64:])
    t1 = torch.nn.functional.elu(ol._ln(x @ spec.get(p2, "t1.W", (64, 64)) + spec.get(p2, "t1.b"), spec.get(p2, "t1.g"), spec.get(p2, "t1.be")))
    t2 = torch.nn.functional.elu(t1 @ spec.get(p2, "t2.W", (64, 32)) + spec.get(p2, "t2.b"))
    t3 = torch.nn.functional.elu(t2 @ spec.get(p2, "t3.W", (32, 16)) + spec.get(p2, "t3.b"))
    exp = t3 @ spec.get(p2, "head.W", (16, 2)) + spec.get(p2, "head.b")
    assert torch.allclose(ol.decode(spec, p2, lat_o, h), exp, atol=1e-13)
