"""Oracle (test infrastructure): the reference SAC update restated on CPU.

Follows, line by line:
  Policy            rl_x/algorithms/sac/flax/policy.py:22-41  (Dense-ReLU-Dense-ReLU -> mean, clipped log_std)
  Critic x2         rl_x/algorithms/sac/flax/critic.py:17-53  (two independent Q nets on [obs, action])
  full-jit nets     rl_x/algorithms/sac/flax_full_jit/policy.py:29-41, critic.py:20-31 (512-LayerNorm-256-128, ELU), key
                    schedule and replay index draw of rl_x/algorithms/sac/flax_full_jit/sac.py:273-282
  EntropyCoefficient rl_x/algorithms/sac/flax/entropy_coefficient.py:5-11  (alpha = exp(log_alpha))
  get_action        rl_x/algorithms/sac/flax/sac.py:119-125
  loss_fn           rl_x/algorithms/sac/flax/sac.py:133-188
  update plumbing   rl_x/algorithms/sac/flax/sac.py:191-215 (per-sample keys, 3 Adam steps, Polyak)
  ReplayBuffer      rl_x/algorithms/sac/flax/replay_buffer.py:4-38 (numpy PCG64 index draws)
PINNED (tests/test_oracle_reference_pin.py) against outputs of the reference's PyTorch flavour executed in the
authoring container (sac/pytorch/{policy,q_network,entropy_coefficient}.py modules, sac.py:90-166 closures; fixture
reference_sac_*.npz): policy_forward, tanh_gaussian, q_forward, loss_and_grads (all three losses and gradients), Adam;
ReplayBuffer bit for bit against the JAX flavour's own numpy class (sac/flax/replay_buffer.py, reference_sac_replay.npz).
sample_noise (per-sample threefry keys): PARITY UNPINNED (JAX only).  The manual backward is also pinned against float64
torch.autograd in tests/test_oracle_sac.py.

FLAT PARAMETER LAYOUT (shared with the HIP library):
  policy : MLP layout of oracle/nets.py with out_dim = 2*A: head columns [0,A) = mean, [A,2A) = raw log_std
  critic : Q0 params followed by Q1 params, each an MLP with in_dim = O + A, out_dim = 1
"""
import math

import numpy as np

from . import nets, prng

LOG_2PI = math.log(2.0 * math.pi)


def make_specs(obs_dim, act_dim, nr_hidden_units=256, arch="flax"):
    """arch "flax": sac/flax/policy.py:22-41, critic.py:17-53 (2 x Dense(H) + ReLU).
    arch "full_jit": sac/flax_full_jit/policy.py:29-41, critic.py:20-31 (Dense(512) + LayerNorm + ELU, Dense(256) + ELU,
    Dense(128) + ELU; the policy's separate mean / log_std Dense heads are the two column blocks of one [128, 2A] head)."""
    if arch == "full_jit":
        ps = nets.MLPSpec(obs_dim, [512, 256, 128], 2 * act_dim, nets.ACT_ELU, True, False)
        qs = nets.MLPSpec(obs_dim + act_dim, [512, 256, 128], 1, nets.ACT_ELU, True, False)
        return ps, qs
    ps = nets.MLPSpec(obs_dim, [nr_hidden_units, nr_hidden_units], 2 * act_dim, nets.ACT_RELU, False, False)
    qs = nets.MLPSpec(obs_dim + act_dim, [nr_hidden_units, nr_hidden_units], 1, nets.ACT_RELU, False, False)
    return ps, qs


def lecun_normal_init(spec, rng, dtype=np.float32):
    """flax Dense default: lecun_normal kernels (truncated normal, std = sqrt(1/fan_in)/.8796), zero biases."""
    p = np.zeros(spec.n_params, dtype=np.float64)
    for L in spec.layers + [spec.head]:
        std = math.sqrt(1.0 / L["in"]) / 0.87962566103423978
        w = np.clip(rng.standard_normal((L["in"], L["out"])), -2, 2) * std
        p[L["W"]:L["W"] + L["in"] * L["out"]] = w.ravel()
        if "g" in L:
            p[L["g"]:L["g"] + L["out"]] = 1.0            # flax LayerNorm: scale 1, bias 0
    return p.astype(dtype)


def sample_noise(key, batch, act_dim, partitionable=True, schedule=0):
    """schedule 0 (sac/flax/sac.py:195-197): keys = split(key, 2B+1); key = keys[0]; eps1[i] = normal(keys[1+2i], (A,));
    eps2[i] = normal(keys[2+2i], (A,)) (per-sample keys through vmap).
    schedule 1 (sac/flax_full_jit/sac.py:273-275): keys = split(key, 2B+2); key = keys[0]; keys[1] samples the replay
    buffer (replay_indices); eps1 uses keys[2 : 2+B], eps2 keys[2+B : 2+2B]."""
    if schedule:
        keys = prng.split(key, 2 * batch + 2, partitionable)
        eps1 = np.stack([prng.normal(keys[2 + i], (act_dim,), partitionable) for i in range(batch)])
        eps2 = np.stack([prng.normal(keys[2 + batch + i], (act_dim,), partitionable) for i in range(batch)])
        return keys[0], eps1, eps2
    keys = prng.split(key, 2 * batch + 1, partitionable)
    eps1 = np.stack([prng.normal(keys[1 + 2 * i], (act_dim,), partitionable) for i in range(batch)])
    eps2 = np.stack([prng.normal(keys[2 + 2 * i], (act_dim,), partitionable) for i in range(batch)])
    return keys[0], eps1, eps2


def replay_indices(key, batch, size, nr_envs, partitionable=True):
    """sac/flax_full_jit/sac.py:273-282: replay key = split(key, 2B+2)[1]; idx1 = randint(rk, (B,), 0, size) and
    idx2 = randint(rk, (B,), 0, nr_envs) -- BOTH from the same key, as the reference has it."""
    rk = prng.split(key, 2 * batch + 2, partitionable)[1]
    return prng.randint(rk, (batch,), 0, size, partitionable), prng.randint(rk, (batch,), 0, nr_envs, partitionable)


def policy_forward(ps, pp, x, log_std_min, log_std_max):
    out, cache = nets.forward(ps, pp, x)
    A = ps.out_dim // 2
    mean, ls_raw = out[:, :A], out[:, A:]
    return mean, np.clip(ls_raw, log_std_min, log_std_max), ls_raw, cache


def q_forward(qs, qp, k, obs, act):
    n = qs.n_params
    out, cache = nets.forward(qs, qp[k * n:(k + 1) * n], np.concatenate([obs, act], axis=1))
    return out[:, 0], cache


def tanh_gaussian(mean, logstd, eps):
    u = mean + np.exp(logstd) * eps
    a = np.tanh(u)
    logp = (-0.5 * ((u - mean) / np.exp(logstd)) ** 2 - 0.5 * LOG_2PI - logstd - np.log(1.0 - a ** 2 + 1e-6)).sum(axis=1)
    return a, logp


def loss_and_grads(ps, pp, qs, qp, qtp, log_alpha, states, next_states, actions, rewards, terminations, eps1, eps2,
                   gamma, target_entropy, log_std_min=-20.0, log_std_max=2.0, critic_states=None, critic_next_states=None,
                   relu_toggle=(), batch_global=None, min_toggle=()):
    """loss_fn (sac.py:133-188) meaned over the batch + manual reverse pass.
    Returns (metrics, gpolicy, gcritic, g_log_alpha).
    relu_toggle: (path, layer, row, unit) entries whose ReLU on/off state is inverted in the REVERSE pass only -- path "q0" / "q1"
    (critics on the replayed action), "qa0" / "qa1" (critics on the policy's action), "pi" (policy on `states`).  For tests that
    bound what an fp32 evaluation may legitimately return when a unit sits within rounding of its kink; forward values untouched.
    min_toggle: rows whose min(Q1, Q2) selection in the policy loss is inverted in the reverse pass only (the same purpose, for a
    row whose two critics agree to within fp32 rounding: d min / da then follows either critic).
    batch_global: the rows are one rank's shard of a batch of that many samples (data parallel): every mean becomes
    sum / batch_global, so gradients and the returned metric SUMS ("sum/q_loss", "sum/min_q", "sum/logp") add up over the ranks to
    the one-device values; the other metrics are then meaningless per rank and are not returned.
    critic_states / critic_next_states: the critics' own observation columns (`x[..., critic_observation_indices]`,
    sac/flax/critic.py:11,23) when they differ from the policy's (states / next_states = policy columns, policy.py:14,31)."""
    cs_ = states if critic_states is None else critic_states
    cs2_ = next_states if critic_next_states is None else critic_next_states
    dt = pp.dtype
    B_rows = states.shape[0]
    B = B_rows if batch_global is None else batch_global          # the divisor of every mean
    A = ps.out_dim // 2
    alpha = np.exp(log_alpha)
    # ---- critic loss
    nm, nls, _, _ = policy_forward(ps, pp, next_states, log_std_min, log_std_max)
    na, nlogp = tanh_gaussian(nm, nls, eps1)
    qt0, _ = q_forward(qs, qtp, 0, cs2_, na)
    qt1, _ = q_forward(qs, qtp, 1, cs2_, na)
    y = rewards + gamma * (1 - terminations) * (np.minimum(qt0, qt1) - alpha * nlogp)
    q0, c0 = q_forward(qs, qp, 0, cs_, actions)
    q1, c1 = q_forward(qs, qp, 1, cs_, actions)
    q_loss = 0.5 * ((q0 - y) ** 2 + (q1 - y) ** 2)          # mean over the 2 critics (q has shape [2,1] per sample)
    n = qs.n_params
    gcritic = np.zeros(2 * n, dtype=dt)

    def toggle(cache, path):
        for pth, layer, row, unit in relu_toggle:
            if pth == path:
                h = cache["h"][layer] = cache["h"][layer].copy()
                h[row, unit] = 0.0 if h[row, unit] > 0 else np.finfo(h.dtype).tiny     # the mask is (h > 0), nets._act_grad
    toggle(c0, "q0")
    toggle(c1, "q1")
    for k, (q, c) in enumerate(((q0, c0), (q1, c1))):
        d = ((q - y) / B)[:, None].astype(dt)               # d mean(q_loss)/dq_k = 2 (q_k - y) / (2 B)
        gcritic[k * n:(k + 1) * n] = nets.backward(qs, qp[k * n:(k + 1) * n], c, d)
    # ---- policy loss
    cm, cls, cls_raw, pc = policy_forward(ps, pp, states, log_std_min, log_std_max)
    std = np.exp(cls)
    u = cm + std * eps2
    ca = np.tanh(u)
    clogp = (-0.5 * ((u - cm) / std) ** 2 - 0.5 * LOG_2PI - cls - np.log(1.0 - ca ** 2 + 1e-6)).sum(axis=1)
    entropy = -clogp
    qa0, ca0 = q_forward(qs, qp, 0, cs_, ca)
    qa1, ca1 = q_forward(qs, qp, 1, cs_, ca)
    min_q = np.minimum(qa0, qa1)
    policy_loss = alpha * clogp - min_q
    toggle(ca0, "qa0")
    toggle(ca1, "qa1")
    toggle(pc, "pi")
    # d(-min_q)/d action through the argmin critic (ties: first)
    sel0 = qa0 <= qa1
    if len(min_toggle):
        sel0 = sel0.copy()
        sel0[list(min_toggle)] ^= True
    O = cs_.shape[1]
    dq_da = np.zeros_like(ca)
    for k, (c, sel) in enumerate(((ca0, sel0), (ca1, ~sel0))):
        d = (np.where(sel, -1.0, 0.0) / B)[:, None].astype(dt)
        _, dx = nets.backward(qs, qp[k * n:(k + 1) * n], c, d, need_dx=True)
        dq_da += dx[:, O:]
    one_m = 1.0 - ca ** 2
    dlogp_da = 2.0 * ca / (one_m + 1e-6)                     # d(-log(1 - a^2 + 1e-6))/da
    d_u = (alpha / B * dlogp_da + dq_da) * one_m             # through a = tanh(u)
    d_mean = d_u
    inside = (cls_raw > log_std_min) & (cls_raw < log_std_max)
    d_ls = np.where(inside, d_u * std * eps2 - alpha / B, 0.0)   # -logstd term: -alpha/B ; u = mean + std*eps
    d_out = np.concatenate([d_mean, d_ls], axis=1).astype(dt)
    gpolicy = nets.backward(ps, pp, pc, d_out)
    # ---- entropy coefficient
    if batch_global is not None:
        return {"sum/q_loss": q_loss.sum(), "sum/min_q": min_q.sum(), "sum/logp": clogp.sum()}, gpolicy, gcritic, None
    entropy_loss = alpha * (entropy - target_entropy)
    g_log_alpha = alpha * (entropy - target_entropy).mean()
    metrics = {"loss/q_loss": q_loss.mean(), "loss/policy_loss": policy_loss.mean(),
               "loss/entropy_loss": entropy_loss.mean(), "entropy/entropy": entropy.mean(), "entropy/alpha": alpha,
               "q_value/q_value": min_q.mean()}
    return metrics, gpolicy, gcritic, dt.type(g_log_alpha)


def loss_torch(ps, pp, qs, qp, qtp, log_alpha, states, next_states, actions, rewards, terminations, eps1, eps2,
               gamma, target_entropy, log_std_min=-20.0, log_std_max=2.0):
    """The same loss through torch.autograd (float64)."""
    import torch
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    P = t(pp).requires_grad_(True)
    Q = t(qp).requires_grad_(True)
    QT = t(qtp)
    LA = t(log_alpha).requires_grad_(True)
    A = ps.out_dim // 2
    n = qs.n_params

    def pol(params, x):
        out = nets.torch_forward(ps, params, x)
        return out[:, :A], torch.clamp(out[:, A:], log_std_min, log_std_max)

    def qf(params, k, s, a):
        return nets.torch_forward(qs, params[k * n:(k + 1) * n], torch.cat([s, a], dim=1))[:, 0]

    def tg(mean, ls, eps):
        u = mean + torch.exp(ls) * eps
        a = torch.tanh(u)
        lp = (-0.5 * ((u - mean) / torch.exp(ls)) ** 2 - 0.5 * LOG_2PI - ls - torch.log(1.0 - a ** 2 + 1e-6)).sum(1)
        return a, lp

    s, s2, a = t(states), t(next_states), t(actions)
    nm, nls = pol(P.detach(), s2)
    na, nlp = tg(nm, nls, t(eps1))
    alpha_g = torch.exp(LA)
    alpha = alpha_g.detach()
    y = t(rewards) + gamma * (1 - t(terminations)) * (torch.minimum(qf(QT, 0, s2, na), qf(QT, 1, s2, na)) - alpha * nlp)
    q_loss = 0.5 * ((qf(Q, 0, s, a) - y) ** 2 + (qf(Q, 1, s, a) - y) ** 2)
    cm, cls = pol(P, s)
    ca, clp = tg(cm, cls, t(eps2))
    entropy = (-clp).detach()
    min_q = torch.minimum(qf(Q.detach(), 0, s, ca), qf(Q.detach(), 1, s, ca))
    loss = (q_loss + alpha * clp - min_q + alpha_g * (entropy - target_entropy)).mean()
    loss.backward()
    return loss.item(), P.grad.numpy(), Q.grad.numpy(), LA.grad.item()


class ReplayBuffer:
    """replay_buffer.py:4-38 verbatim semantics (ring of capacity // nr_envs rows, numpy Generator draws)."""

    def __init__(self, capacity, nr_envs, obs_dim, act_dim, rng):
        self.capacity = capacity // nr_envs
        self.nr_envs = nr_envs
        self.rng = rng
        self.states = np.zeros((self.capacity, nr_envs, obs_dim), np.float32)
        self.next_states = np.zeros((self.capacity, nr_envs, obs_dim), np.float32)
        self.actions = np.zeros((self.capacity, nr_envs, act_dim), np.float32)
        self.rewards = np.zeros((self.capacity, nr_envs), np.float32)
        self.terminations = np.zeros((self.capacity, nr_envs), np.float32)
        self.pos = 0
        self.size = 0

    def add(self, states, next_states, actions, rewards, terminations):
        self.states[self.pos] = states
        self.next_states[self.pos] = next_states
        self.actions[self.pos] = actions
        self.rewards[self.pos] = rewards
        self.terminations[self.pos] = terminations
        self.pos = (self.pos + 1) % self.capacity
        self.size = min(self.size + 1, self.capacity)

    def sample_indices(self, nr_samples):
        idx1 = self.rng.integers(self.size, size=nr_samples)
        idx2 = self.rng.integers(self.nr_envs, size=nr_samples)
        return idx1, idx2

    def gather(self, idx1, idx2):
        return (self.states[idx1, idx2], self.next_states[idx1, idx2], self.actions[idx1, idx2],
                self.rewards[idx1, idx2], self.terminations[idx1, idx2])


def processed_action(action, low, high):
    """`get_processed_action`, sac/flax/policy.py:44-48: clip to [-1, 1], rescale to the env's action bounds."""
    c = np.clip(action, -1, 1)
    return low + 0.5 * (c + 1.0) * (high - low)


def polyak(params, target, tau):
    """optax.incremental_update(new, old, tau) = tau*new + (1-tau)*old (sac.py:208)."""
    return (tau * params + (1 - tau) * target).astype(params.dtype)
