"""Oracle (test infrastructure): PPO with a Categorical policy head -- BASELINE.json configs[0] (CartPole).

The reference's JAX flavours have no discrete policy; its only discrete PPO head is
    DiscreteFlatValuesPolicy        rl_x/algorithms/ppo/pytorch/policy.py:96-135
        trunk Dense-tanh-Dense-tanh (oracle.nets arch "A"), logits head orthogonal(0.01), Categorical(logits):
        sample / log_prob / entropy (:118-130), deterministic action = argmax (:133-135)
    loss                            rl_x/algorithms/ppo/pytorch/ppo.py:121-150 (same clipped surrogate; entropy = mean of the
                                    per-sample Categorical entropies)
PINNED (tests/test_oracle_reference_pin.py) against that module and the `policy_loss_fn` closure executed in the authoring
container (fixture reference_ppo_discrete_*.npz): log-probs, entropies, loss terms, gradients, clip + Adam.
sample_categorical restates jax.random.categorical (argmax(logits + Gumbel), the noise construction the HIP acting kernel
uses on the threefry stream): PARITY UNPINNED (JAX only; the PyTorch flavour samples with torch's generator).
"""
import numpy as np

from . import nets, prng


def log_softmax(z):
    m = z.max(axis=1, keepdims=True)
    return z - (m + np.log(np.exp(z - m).sum(axis=1, keepdims=True)))


def categorical_logp_entropy(logits, actions):
    """log_prob of the integer actions and the per-sample entropy (policy.py:126-130)."""
    ls = log_softmax(logits)
    idx = np.asarray(actions).astype(np.int64).reshape(-1)
    return ls[np.arange(ls.shape[0]), idx], -(np.exp(ls) * ls).sum(axis=1)


def sample_categorical(key, logits, partitionable=True):
    """jax.random.categorical(key, logits, axis=-1): argmax(logits + gumbel(key, logits.shape)),
    gumbel = -log(-log(uniform(key, shape, minval=tiny, maxval=1)))."""
    tiny = np.finfo(np.float32).tiny
    u = prng.uniform(key, logits.shape, tiny, 1.0, partitionable)
    g = -np.log(-np.log(u.astype(np.float32))).astype(np.float32)
    return np.argmax(logits.astype(np.float32) + g, axis=1)


def ppo_loss_and_grads(pspec, pparams, cspec, cparams, states, actions, logp_old, returns, adv, clip_range, entropy_coef,
                       critic_coef):
    """Same loss as oracle.ppo.ppo_loss_and_grads with the Categorical policy; adv is already normalised.
    Returns (loss, metrics, gpolicy, gcritic); manual reverse pass (what the HIP head kernel mirrors)."""
    dt = pparams.dtype
    mb = states.shape[0]
    logits, pcache = nets.forward(pspec, pparams, states)
    ls = log_softmax(logits)
    p = np.exp(ls)
    idx = np.asarray(actions).astype(np.int64).reshape(-1)
    onehot = np.zeros_like(ls)
    onehot[np.arange(mb), idx] = 1
    new_logp = (ls * onehot).sum(axis=1)
    ent = -(p * ls).sum(axis=1)
    logratio = new_logp - logp_old
    ratio = np.exp(logratio)
    pg1 = -adv * ratio
    pg2 = -adv * np.clip(ratio, 1 - clip_range, 1 + clip_range)
    pg = np.maximum(pg1, pg2)
    value, ccache = nets.forward(cspec, cparams, states)
    value = value.reshape(-1)
    vl = 0.5 * (value - returns) ** 2
    loss = (pg - entropy_coef * ent + critic_coef * vl).mean()
    metrics = {"loss/policy_gradient_loss": pg.mean(), "loss/critic_loss": vl.mean(), "loss/entropy_loss": ent.mean(),
               "policy_ratio/approx_kl": ((ratio - 1) - logratio).mean(),
               "policy_ratio/clip_fraction": (np.abs(ratio - 1) > clip_range).astype(dt).mean()}
    inside = (ratio >= 1 - clip_range) & (ratio <= 1 + clip_range)
    d_ratio = np.where(inside | (pg1 > pg2), -adv, 0).astype(dt)
    d_logp = d_ratio * ratio / mb
    # d logp / d z_j = [j == a] - p_j ;  d(-c * mean H) / d z_j = c / mb * p_j (log p_j + H)
    d_logits = d_logp[:, None] * (onehot - p) + (entropy_coef / mb) * p * (ls + ent[:, None])
    gpol = nets.backward(pspec, pparams, pcache, d_logits.astype(dt))
    d_v = (critic_coef / mb) * (value - returns)
    gcrit = nets.backward(cspec, cparams, ccache, d_v[:, None].astype(dt))
    return loss, metrics, gpol, gcrit
