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
from oracle import env as oenv, nets, ppo as oppo, prng  # noqa: E402

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


if __name__ == "__main__":
    prng_vectors()
    gae_case()
    minibatch_case("A")
    minibatch_case("B")
    env_case()
    print("golden fixtures written to", HERE)
