#!/usr/bin/env python
"""Generates tests/golden/*.npz from the CPU oracle (source: "restatement" -- the reference's
own Python cannot be imported in the authoring container: jax / flax / optax are absent,
SURVEY.md F4; regenerate with source "reference-import" if a JAX-capable host ever exists).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import env as oenv, nets, ppo as oppo, ppo_lstm as olstm, prng, sac as osac  # noqa: E402

SOURCE = "restatement"


def prng_vectors():
    out = {"source": SOURCE}
    for scheme in (0, 1):
        p = bool(scheme)
        key = prng.prng_key(1)
        out[f"split3_s{scheme}"] = prng.split(key, 3, p)
        out[f"bits1001_s{scheme}"] = prng.random_bits(prng.prng_key(1234), (1001,), p)
        out[f"normal_s{scheme}"] = prng.normal(prng.prng_key(7), (512,), p)
        k2, idx = prng.ppo_minibatch_indices(prng.prng_key(1), 8 * 64, 3, 4, 128, p)
        out[f"perm_key_s{scheme}"] = k2
        out[f"perm_idx_s{scheme}"] = idx
    np.savez_compressed(os.path.join(HERE, "prng.npz"), **out)


def gae_case():
    rng = np.random.default_rng(11)
    T, N = 16, 48
    r = rng.standard_normal((T, N)).astype(np.float32)
    v = rng.standard_normal((T, N)).astype(np.float32)
    nv = rng.standard_normal((T, N)).astype(np.float32)
    term = (rng.random((T, N)) < 0.15).astype(np.float32)
    adv, ret = oppo.gae(r.astype(np.float64), v.astype(np.float64), nv.astype(np.float64), term.astype(np.float64), 0.99, 0.9)
    np.savez_compressed(os.path.join(HERE, "gae.npz"), source=SOURCE, rewards=r, values=v, next_values=nv,
                        terminations=term, gamma=0.99, gae_lambda=0.9, advantages=adv, returns=ret)


def minibatch_case(arch):
    rng = np.random.default_rng(5 if arch == "B" else 6)
    O, A, B, mb = 17, 6, 512, 192
    ps, cs = nets.make_spec(arch, O, A, True), nets.make_spec(arch, O, 1, False)
    pp = (nets.init_params(ps, rng, 0.01) + 0.05 * rng.standard_normal(ps.n_params)).astype(np.float32)
    cp = (nets.init_params(cs, rng, 1.0) + 0.05 * rng.standard_normal(cs.n_params)).astype(np.float32)
    states = rng.standard_normal((B, O)).astype(np.float32)
    actions = rng.standard_normal((B, A)).astype(np.float32)
    mean, _ = nets.forward(ps, pp, states)
    logp = (oppo.gaussian_log_prob(actions, mean, pp[ps.logstd:][None, :]) + 0.05 * rng.standard_normal(B)).astype(np.float32)
    returns = rng.standard_normal(B).astype(np.float32)
    adv = (2 * rng.standard_normal(B) + 0.3).astype(np.float32)
    idx = rng.permutation(B)[:mb].astype(np.int32)
    f64 = lambda a: a.astype(np.float64)
    madv = oppo.normalize_advantages(f64(adv[idx]))
    loss, met, gp, gc = oppo.ppo_loss_and_grads(ps, f64(pp), cs, f64(cp), f64(states[idx]), f64(actions[idx]),
                                                f64(logp[idx]), f64(returns[idx]), madv, 0.1, 0.01, 0.7)
    # one optimizer step from zero moments (optax chain(clip, adam))
    pst, cst = oppo.TrainState(ps, f64(pp)), oppo.TrainState(cs, f64(cp))
    pn = pst.apply_gradients(gp, 4e-4, 0.5)
    cn = cst.apply_gradients(gc, 4e-4, 0.5)
    np.savez_compressed(os.path.join(HERE, f"minibatch_{arch}.npz"), source=SOURCE, arch=arch, obs_dim=O, act_dim=A,
                        pparams=pp, cparams=cp, states=states, actions=actions, log_probs=logp, returns=returns,
                        advantages=adv, idx=idx, clip_range=0.1, entropy_coef=0.01, critic_coef=0.7,
                        loss=loss, pg_loss=met["loss/policy_gradient_loss"], critic_loss=met["loss/critic_loss"],
                        entropy_loss=met["loss/entropy_loss"], approx_kl=met["policy_ratio/approx_kl"],
                        clip_fraction=met["policy_ratio/clip_fraction"], pgrads=gp.astype(np.float32), cgrads=gc.astype(np.float32),
                        policy_grad_norm=pn, critic_grad_norm=cn, pparams_after=pst.params.astype(np.float32), cparams_after=cst.params.astype(np.float32),
                        lr=4e-4, max_grad_norm=0.5)


def env_case():
    N, O, A = 96, 17, 6
    o = oenv.RandomObsEnvOracle(3, N, O, A, horizon=12, p_term=0.05, reward_noise=0.1, env_id_offset=32)
    obs0 = o.reset()
    rng = np.random.default_rng(0)
    acts, obs, fin, rew, term, trunc = [], [], [], [], [], []
    for _ in range(20):
        a = rng.standard_normal((N, A)).astype(np.float32)
        ob, fi, r, te, tr, _ = o.step(a)
        acts.append(a); obs.append(ob); fin.append(fi); rew.append(r); term.append(te); trunc.append(tr)
    np.savez_compressed(os.path.join(HERE, "env.npz"), source=SOURCE, seed=3, N=N, O=O, A=A, horizon=12, p_term=0.05,
                        reward_noise=0.1, env_id_offset=32, obs0=obs0, actions=np.stack(acts), obs=np.stack(obs),
                        final_obs=np.stack(fin), rewards=np.stack(rew), terminated=np.stack(term),
                        truncated=np.stack(trunc))


def sac_case():
    """One SAC update (sac/flax/sac.py:128-215) at a small Humanoid-like shape: inputs, per-sample noise, losses, gradients."""
    rng = np.random.default_rng(21)
    O, A, B, H = 40, 6, 96, 64
    ps, qs = osac.make_specs(O, A, H)
    pp = (osac.lecun_normal_init(ps, rng) + 0.02 * rng.standard_normal(ps.n_params)).astype(np.float32)
    pp[ps.head["W"]:ps.head["W"] + ps.head["in"] * ps.head["out"]] *= 0.1
    qp = (np.concatenate([osac.lecun_normal_init(qs, rng) for _ in range(2)]) + 0.02 * rng.standard_normal(2 * qs.n_params)).astype(np.float32)
    qtp = (qp + 0.01 * rng.standard_normal(qp.shape)).astype(np.float32)
    s = rng.standard_normal((B, O)).astype(np.float32)
    s2 = rng.standard_normal((B, O)).astype(np.float32)
    a = np.tanh(rng.standard_normal((B, A))).astype(np.float32)
    r = rng.standard_normal(B).astype(np.float32)
    term = (rng.random(B) < 0.2).astype(np.float32)
    log_alpha = np.float32(-0.3)
    key = prng.prng_key(11)
    f = lambda x: x.astype(np.float64)
    new_key, e1, e2 = osac.sample_noise(key, B, A, True)
    met, gp, gq, ga = osac.loss_and_grads(ps, f(pp), qs, f(qp), f(qtp), np.float64(log_alpha), f(s), f(s2), f(a), f(r), f(term),
                                          f(e1), f(e2), 0.99, -float(A))
    np.savez_compressed(os.path.join(HERE, "sac.npz"), source=SOURCE, obs_dim=O, act_dim=A, hidden=H, pparams=pp, qparams=qp,
                        qtarget=qtp, log_alpha=log_alpha, states=s, next_states=s2, actions=a, rewards=r, terminations=term,
                        key=key, new_key=new_key, eps_next=e1.astype(np.float32), eps_cur=e2.astype(np.float32), gamma=0.99,
                        target_entropy=-float(A), q_loss=met["loss/q_loss"], policy_loss=met["loss/policy_loss"],
                        entropy_loss=met["loss/entropy_loss"], entropy=met["entropy/entropy"], alpha=met["entropy/alpha"],
                        q_value=met["q_value/q_value"], gpolicy=gp.astype(np.float32), gcritic=gq.astype(np.float32),
                        g_log_alpha=np.float32(ga))


def lstm_case():
    """One PPO+LSTM sequence minibatch (ppo_lstm.py:181-259): env-index permutation, loss terms, BPTT gradients (float64
    torch autograd through the restated policy)."""
    import torch
    rng = np.random.default_rng(31)
    T, N, ne, O, A = 6, 24, 16, 17, 6
    spec = olstm.LstmPolicySpec(O, A, 128, 64, (512, 256, 128), False)
    p = (olstm.init_params(spec, rng, 1.0) + 0.03 * rng.standard_normal(spec.n_params)).astype(np.float32)
    cs = nets.make_spec("B", O, 1, False)
    cp = (nets.init_params(cs, rng, 1.0) + 0.03 * rng.standard_normal(cs.n_params)).astype(np.float32)
    states = rng.standard_normal((T, N, O)).astype(np.float32)
    actions = rng.standard_normal((T, N, A)).astype(np.float32)
    dones = (rng.random((T, N)) < 0.2).astype(np.float32)
    c0 = (0.5 * rng.standard_normal((N, 64))).astype(np.float32)
    h0 = np.tanh(0.5 * rng.standard_normal((N, 64))).astype(np.float32)
    t64 = lambda x: torch.tensor(np.asarray(x, dtype=np.float64))
    with torch.no_grad():
        mean = olstm.forward_sequence(spec, t64(p), t64(states), t64(dones), t64(c0), t64(h0)).numpy()
    logstd = p[spec.off["logstd"][0]:][:A].astype(np.float64)[None, None, :]
    logp = ((-0.5 * ((actions - mean) / np.exp(logstd)) ** 2 - 0.5 * olstm.LOG_2PI - logstd).sum(-1)
            + 0.05 * rng.standard_normal((T, N))).astype(np.float32)
    returns = rng.standard_normal((T, N)).astype(np.float32)
    adv = (2 * rng.standard_normal((T, N)) + 0.3).astype(np.float32)
    key = prng.prng_key(5)
    new_key, idx = olstm.env_minibatch_indices(key, N, 2, N // 8, 8)
    env_idx = rng.permutation(N)[:ne].astype(np.int32)
    P, C = t64(p).requires_grad_(True), t64(cp).requires_grad_(True)
    a_n = adv[:, env_idx].astype(np.float64)
    a_n = (a_n - a_n.mean()) / (a_n.std() + 1e-8)
    loss, met = olstm.ppo_lstm_loss(spec, P, cs, C, t64(states[:, env_idx]), t64(actions[:, env_idx]), t64(logp[:, env_idx]),
                                    t64(returns[:, env_idx]), t64(a_n), t64(dones[:, env_idx]), t64(c0[env_idx]), t64(h0[env_idx]),
                                    0.1, 0.01, 0.7)
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "ppo_lstm.npz"), source=SOURCE, obs_dim=O, act_dim=A, pparams=p, cparams=cp,
                        states=states, actions=actions, log_probs=logp, returns=returns, advantages=adv, dones=dones, c0=c0, h0=h0,
                        env_idx=env_idx, clip_range=0.1, entropy_coef=0.01, critic_coef=0.7, key=key, perm_key=new_key,
                        perm_env_idx=idx, loss=float(loss), pg_loss=float(met["loss/policy_gradient_loss"]),
                        critic_loss=float(met["loss/critic_loss"]), entropy_loss=float(met["loss/entropy_loss"]),
                        approx_kl=float(met["policy_ratio/approx_kl"]), clip_fraction=float(met["policy_ratio/clip_fraction"]),
                        pgrads=P.grad.numpy().astype(np.float32), cgrads=C.grad.numpy().astype(np.float32))


if __name__ == "__main__":
    sac_case()
    lstm_case()
    prng_vectors()
    gae_case()
    minibatch_case("A")
    minibatch_case("B")
    env_case()
    print("golden fixtures written to", HERE)
