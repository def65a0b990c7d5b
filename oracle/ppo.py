"""Oracle (test infrastructure): the reference PPO iteration restated on CPU.

Follows, line by line:
  sampling / log-prob      rl_x/algorithms/ppo/flax/ppo.py:110-119
                           (full-jit twin: ppo/flax_full_jit/ppo.py:133-140)
  GAE                      rl_x/algorithms/ppo/flax/ppo.py:122-135
  loss_fn                  rl_x/algorithms/ppo/flax/ppo.py:142-177
  minibatch permutation    rl_x/algorithms/ppo/flax/ppo.py:191-194
  minibatch_update         rl_x/algorithms/ppo/flax/ppo.py:196-220
  optimizer                rl_x/algorithms/ppo/flax/ppo.py:76-100 (optax semantics
                           restated: optax>=0.2.6 is not under /root/reference)
  metrics                  rl_x/algorithms/ppo/flax/ppo.py:226-230
PINNED (tests/test_oracle_reference_pin.py) against outputs of the reference's PyTorch flavour executed in the
authoring container (ppo/pytorch/ppo.py:98-166 closures, policy.py, critic.py; fixtures reference_ppo_*.npz):
gaussian_log_prob, processed_action, gae, ppo_loss_and_grads, clip_by_global_norm + adam_step.  The JAX flavour
differs in `jnp.std` (population) vs `Tensor.std()` (unbiased) -- normalize_advantages below is the JAX form and is
NOT what the fixture used -- and in optax's clip (c/norm vs c/(norm+1e-6)).  `update`'s permutation / key schedule:
PARITY UNPINNED (JAX only).  Also pinned by closed forms and float64 torch.autograd in tests/test_oracle_ppo.py.
"""
import math
import numpy as np

from . import nets, prng

LOG_2PI = math.log(2.0 * math.pi)


# ----------------------------------------------------------------- acting
def processed_action(action, clip_and_rescale, low, high):
    """ppo/flax/policy.py:43-50."""
    if not clip_and_rescale:
        return action
    c = np.clip(action, -1, 1)
    return (low + 0.5 * (c + 1.0) * (high - low)).astype(action.dtype)


def gaussian_log_prob(action, mean, logstd):
    std = np.exp(logstd)
    return (-0.5 * ((action - mean) / std) ** 2 - 0.5 * action.dtype.type(LOG_2PI) - logstd).sum(axis=1)


def get_action_and_value(pspec, pparams, cspec, cparams, state, noise):
    """ppo/flax/ppo.py:110-119 with the N(0,1) draw `noise` supplied by the caller."""
    mean, _ = nets.forward(pspec, pparams, state)
    logstd = pparams[pspec.logstd:pspec.logstd + pspec.out_dim][None, :]
    action = (mean + np.exp(logstd) * noise).astype(pparams.dtype)
    logp = gaussian_log_prob(action, mean, logstd)
    value, _ = nets.forward(cspec, cparams, state)
    return action, value.reshape(-1), logp


# -------------------------------------------------------------------- GAE
def gae(rewards, values, next_values, terminations, gamma, lam):
    """ppo/flax/ppo.py:124-134.  All [T,N].  Masks with terminations only."""
    dt = rewards.dtype
    g, l = dt.type(gamma), dt.type(lam)
    one = dt.type(1)
    delta = rewards + g * next_values * (one - terminations) - values
    T = rewards.shape[0]
    adv = np.zeros_like(rewards)
    adv[T - 1] = delta[T - 1]
    for t in range(T - 2, -1, -1):
        adv[t] = delta[t] + g * l * (one - terminations[t]) * adv[t + 1]
    return adv, adv + values


# ------------------------------------------------------------------- loss
def normalize_advantages(a):
    """ppo/flax/ppo.py:199-200 (population std)."""
    dt = a.dtype
    return ((a - a.mean()) / (a.std() + dt.type(1e-8))).astype(dt)


def ppo_loss_and_grads(pspec, pparams, cspec, cparams, states, actions, logp_old, returns, adv,
                       clip_range, entropy_coef, critic_coef, critic_states=None):
    """loss_fn (ppo.py:142-177) meaned over the minibatch (:186-188) + the manual
    reverse pass the HIP kernels mirror.  Returns (loss, metrics dict, gpol, gcrit).
    critic_states: the critic's own observation columns `x[..., critic_observation_indices]` (ppo/flax/critic.py:12,24) when
    they differ from the policy's (`states` then = `x[..., policy_observation_indices]`, policy.py:13,33)."""
    dt = pparams.dtype
    mb = states.shape[0]
    A = pspec.out_dim
    mean, pcache = nets.forward(pspec, pparams, states)
    logstd = pparams[pspec.logstd:pspec.logstd + A][None, :]
    std = np.exp(logstd)
    zs = (actions - mean) / std
    new_logp = (-0.5 * zs ** 2 - dt.type(0.5 * LOG_2PI) - logstd).sum(axis=1)
    entropy = (logstd + dt.type(0.5 * math.log(2.0 * math.pi * math.e))).sum()
    logratio = new_logp - logp_old
    ratio = np.exp(logratio)
    approx_kl = (ratio - 1) - logratio
    clip_frac = (np.abs(ratio - 1) > clip_range).astype(dt)
    pg1 = -adv * ratio
    pg2 = -adv * np.clip(ratio, 1 - clip_range, 1 + clip_range)
    pg = np.maximum(pg1, pg2)
    value, ccache = nets.forward(cspec, cparams, states if critic_states is None else critic_states)
    value = value.reshape(-1)
    vl = 0.5 * (value - returns) ** 2
    loss = (pg - entropy_coef * entropy + critic_coef * vl).mean()
    metrics = {
        "loss/policy_gradient_loss": pg.mean(), "loss/critic_loss": vl.mean(),
        "loss/entropy_loss": entropy, "policy_ratio/approx_kl": approx_kl.mean(),
        "policy_ratio/clip_fraction": clip_frac.mean(),
    }
    # ---- manual backward
    inside = (ratio >= 1 - clip_range) & (ratio <= 1 + clip_range)
    d_ratio = np.where(inside | (pg1 > pg2), -adv, 0).astype(dt)
    d_logp = d_ratio * ratio / mb                                    # [mb]
    d_mean = d_logp[:, None] * (zs / std)                            # dlogp/dmean = (a-mu)/sigma^2
    gpol = nets.backward(pspec, pparams, pcache, d_mean.astype(dt))
    d_logstd = (d_logp[:, None] * (zs ** 2 - 1)).sum(axis=0) - entropy_coef
    gpol[pspec.logstd:pspec.logstd + A] = d_logstd
    d_v = (critic_coef / mb) * (value - returns)
    gcrit = nets.backward(cspec, cparams, ccache, d_v[:, None].astype(dt))
    return loss, metrics, gpol, gcrit


def ppo_loss_torch(pspec, pparams, cspec, cparams, states, actions, logp_old, returns, adv,
                   clip_range, entropy_coef, critic_coef, dtype=None, device=None):
    """The same loss through torch.autograd (independent check of the manual backward).  device: where to evaluate it
    (default CPU; the full-size GPU tests pass the HIP device for a float64 evaluation of a 32768-row minibatch in < 1 s)."""
    import torch
    dtype = dtype or torch.float64
    t = lambda a: torch.tensor(np.asarray(a), dtype=dtype, device=device)
    pp = t(pparams).requires_grad_(True)
    cp = t(cparams).requires_grad_(True)
    A = pspec.out_dim
    mean = nets.torch_forward(pspec, pp, t(states))
    logstd = pp[pspec.logstd:pspec.logstd + A][None, :]
    std = torch.exp(logstd)
    new_logp = (-0.5 * ((t(actions) - mean) / std) ** 2 - 0.5 * LOG_2PI - logstd).sum(1)
    entropy = (logstd + 0.5 * math.log(2.0 * math.pi * math.e)).sum(1)
    logratio = new_logp - t(logp_old)
    ratio = torch.exp(logratio)
    pg = torch.maximum(-t(adv) * ratio, -t(adv) * torch.clamp(ratio, 1 - clip_range, 1 + clip_range))
    value = nets.torch_forward(cspec, cp, t(states)).reshape(-1)
    vl = 0.5 * (value - t(returns)) ** 2
    loss = (pg - entropy_coef * entropy + critic_coef * vl).mean()
    loss.backward()
    return loss.item(), pp.grad.cpu().numpy(), cp.grad.cpu().numpy()


# -------------------------------------------------------------- optimizer
def global_norm(g):
    return np.sqrt((g.astype(np.float64) ** 2).sum()).astype(g.dtype) if g.dtype == np.float64 else \
        np.float32(np.sqrt(np.float32((g * g).sum(dtype=np.float32))))


def clip_by_global_norm(g, max_norm):
    """optax.clip_by_global_norm: g if ||g|| < c else (g/||g||)*c."""
    n = global_norm(g)
    if n < max_norm:
        return g, n
    return ((g / n) * g.dtype.type(max_norm)).astype(g.dtype), n


def adam_step(p, g, m, v, count, lr, b1=0.9, b2=0.999, eps=1e-8):
    """optax.adam (eps_root=0): count is the number of PREVIOUS steps."""
    dt = p.dtype
    b1, b2 = dt.type(b1), dt.type(b2)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    t = count + 1
    mhat = m / dt.type(1 - float(b1) ** t)
    vhat = v / dt.type(1 - float(b2) ** t)
    p = p - dt.type(lr) * (mhat / (np.sqrt(vhat) + dt.type(eps)))
    return p.astype(dt), m.astype(dt), v.astype(dt)


def linear_schedule(lr0, count, nr_minibatches, nr_epochs, nr_updates):
    """ppo/flax/ppo.py:76-80."""
    return lr0 * (1.0 - (count // (nr_minibatches * nr_epochs)) / nr_updates)


class TrainState:
    def __init__(self, spec, params):
        self.spec = spec
        self.params = params.copy()
        self.m = np.zeros_like(params)
        self.v = np.zeros_like(params)
        self.count = 0

    def apply_gradients(self, g, lr, max_grad_norm):
        gc, n = clip_by_global_norm(g, max_grad_norm)
        self.params, self.m, self.v = adam_step(self.params, gc, self.m, self.v, self.count, lr)
        self.count += 1
        return n


# ------------------------------------------------------------- one update
def update(pstate, cstate, states, actions, advantages, returns, values, log_probs, key, cfg,
           partitionable=True, batch_indices=None):
    """`update` (ppo/flax/ppo.py:138-232).  Arrays are time-major [T,N,...].
    Returns (per-update metric lists, new key, batch_indices)."""
    O = states.shape[-1]
    A = actions.shape[-1]
    bs = states.reshape(-1, O)
    ba = actions.reshape(-1, A)
    badv = advantages.reshape(-1)
    bret = returns.reshape(-1)
    blp = log_probs.reshape(-1)
    B = bs.shape[0]
    M = B // cfg["minibatch_size"]
    E = cfg["nr_epochs"]
    if batch_indices is None:
        key, batch_indices = prng.ppo_minibatch_indices(key, B, E, M, cfg["minibatch_size"], partitionable)
    else:
        key = prng.split(key, 2, partitionable)[0]
    out = []
    for idx in batch_indices:
        madv = normalize_advantages(badv[idx])
        lr = cfg["learning_rate"]
        if cfg.get("anneal_learning_rate", False):
            lr = linear_schedule(cfg["learning_rate"], pstate.count, M, E, cfg["nr_updates"])
        loss, metrics, gp, gc = ppo_loss_and_grads(
            pstate.spec, pstate.params, cstate.spec, cstate.params,
            bs[idx], ba[idx], blp[idx], bret[idx], madv,
            cfg["clip_range"], cfg["entropy_coef"], cfg["critic_coef"])
        metrics["gradients/policy_grad_norm"] = pstate.apply_gradients(gp, lr, cfg["max_grad_norm"])
        metrics["gradients/critic_grad_norm"] = cstate.apply_gradients(gc, lr, cfg["max_grad_norm"])
        metrics["loss"] = loss
        metrics["lr/learning_rate"] = lr
        out.append(metrics)
    return out, key, batch_indices


def explained_variance(returns, values):
    return 1 - np.var(returns - values) / (np.var(returns) + 1e-8)
