"""CPU restatement of FastSAC's networks and update steps -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Follows rl_x/algorithms/fastsac/pytorch:
    policy.py:46-57     torso Linear(O,512)-LayerNorm-SiLU, Linear(512,256)-LayerNorm-SiLU, Linear(256,128)-LayerNorm-SiLU
                        (torch.nn.LayerNorm: population variance, eps 1e-5), heads mean / log_std Linear(128, A)
    policy.py:66-72     log_std = min + 0.5 (max - min) (tanh(raw) + 1)
    policy.py:75-90     a = tanh(mean + std eps) * action_scale;  log_prob = sum_j Normal(mean, std).log_prob(raw)
                        - log(1 - tanh^2 + 1e-6) - log(action_scale + 1e-6)
    policy.py:36-43     action_scale = max(|low - center|, |high - center|) / scale
    q_network.py:27-43  critic Linear(Oc + A,768)-LN-SiLU, Linear(768,384)-LN-SiLU, Linear(384,192)-LN-SiLU, Linear(192, nr_atoms)
    fastsac.py:144-241  critic_and_entropy_loss_fn: next action / log-prob from the policy (no gradient), categorical projection
                        of the entropy-adjusted n-step target (oracle/c51.py restates it), q_loss = q1_loss + q2_loss,
                        AdamW step of both critics; entropy_loss = mean(exp(log_alpha) (entropy - target_entropy)) with
                        entropy = -next_log_probs, AdamW step of log_alpha
    fastsac.py:323-327  Polyak: target = (1 - tau) target + tau param, every parameter of both critics
    fastsac.py:106-141  policy_loss_fn: loss = mean(alpha log_prob - q) with q = (q1 + q2) / 2 (or min) of the EXPECTED values
                        sum_j softmax(logits)_j z_j; AdamW step of the policy
    fastsac.py:88-91    torch.optim.AdamW(lr, weight_decay, betas): p *= 1 - lr wd; m, v moments; p -= lr / (1 - b1^t) * m /
                        (sqrt(v) / sqrt(1 - b2^t) + 1e-8)

Parameters live in the library's flat layout (include/rlx_hip.h, rlx_lnmlp_desc): per hidden layer W[in, out] row-major, b,
LayerNorm scale, LayerNorm bias; then the head W[in, out], b.  The policy's head is [mean | log_std], 2A wide.

Pinned by tests/golden/reference_fastsac.npz: outputs of the reference's own modules and closures executed in float64
(tests/golden/make_reference_golden.py: make_fastsac), checked in tests/test_oracle_fastsac.py."""
import math

import numpy as np
import torch

from oracle import c51

POLICY_HIDDEN = (512, 256, 128)
CRITIC_HIDDEN = (768, 384, 192)
LN_EPS = 1e-5


def param_count(in_dim, hidden, out_dim):
    n, d = 0, in_dim
    for h in hidden:
        n += d * h + 3 * h
        d = h
    return n + d * out_dim + out_dim


def blocks(in_dim, hidden, out_dim):
    """[(name, offset, size)] of the flat layout."""
    out, off, d = [], 0, in_dim
    for li, h in enumerate(hidden):
        for name, n in (("W%d" % li, d * h), ("b%d" % li, h), ("g%d" % li, h), ("be%d" % li, h)):
            out.append((name, off, n))
            off += n
        d = h
    out += [("Wh", off, d * out_dim), ("bh", off + d * out_dim, out_dim)]
    return out


def make_params(seed, obs_dim, act_dim, nr_atoms, critic_obs_dim=None):
    """Deterministic test parameters in the flat layout: (policy, [q1, q2, q1_target, q2_target]) float32 arrays drawn from
    numpy's PCG64 -- the fixture generator loads exactly these into the reference's modules, a test rebuilds them from the seed."""
    rng = np.random.default_rng(seed)
    oc = obs_dim if critic_obs_dim is None else critic_obs_dim

    def net(in_dim, hidden, out_dim, head_scale):
        parts, d = [], in_dim
        for h in hidden:
            parts += [rng.standard_normal((d, h)) / np.sqrt(d), 0.1 * rng.standard_normal(h), 1.0 + 0.1 * rng.standard_normal(h),
                      0.1 * rng.standard_normal(h)]
            d = h
        parts += [head_scale * rng.standard_normal((d, out_dim)) / np.sqrt(d), 0.1 * rng.standard_normal(out_dim)]
        return np.concatenate([x.reshape(-1) for x in parts]).astype(np.float32)
    policy = net(obs_dim, POLICY_HIDDEN, 2 * act_dim, 0.5)
    critics = [net(oc + act_dim, CRITIC_HIDDEN, nr_atoms, 1.0) for _ in range(4)]
    return policy, critics


def forward(flat, in_dim, hidden, out_dim, x, keep=None):
    """flat, x: torch tensors (any float dtype).  -> head output [M, out_dim].  keep (a list) receives (z_l, h_l) per layer."""
    off, d, h = 0, in_dim, x
    for width in hidden:
        W = flat[off:off + d * width].reshape(d, width); off += d * width
        b = flat[off:off + width]; off += width
        g = flat[off:off + width]; off += width
        be = flat[off:off + width]; off += width
        z = h @ W + b
        mean = z.mean(dim=1, keepdim=True)
        var = ((z - mean) ** 2).mean(dim=1, keepdim=True)
        y = (z - mean) / torch.sqrt(var + LN_EPS) * g + be
        h = y * torch.sigmoid(y)
        if keep is not None:
            keep.append((z, h))
        d = width
    W = flat[off:off + d * out_dim].reshape(d, out_dim); off += d * out_dim
    b = flat[off:off + out_dim]
    return h @ W + b


def policy_forward(pflat, obs_dim, act_dim, obs, log_std_min, log_std_max):
    head = forward(pflat, obs_dim, POLICY_HIDDEN, 2 * act_dim, obs)
    mean, raw = head[:, :act_dim], head[:, act_dim:]
    log_std = log_std_min + 0.5 * (log_std_max - log_std_min) * (torch.tanh(raw) + 1.0)
    return mean, log_std


def sample(mean, log_std, eps, action_scale):
    """-> (scaled action, log-prob) as policy.get_action_and_log_prob computes them for the noise eps."""
    std = torch.exp(log_std)
    raw = mean + std * eps
    t = torch.tanh(raw)
    logp = -((raw - mean) ** 2) / (2.0 * std * std) - log_std - math.log(math.sqrt(2.0 * math.pi))
    # (policy.py:36-43 builds action_scale from float32 tensors and it is no module buffer: it STAYS float32 whatever the module's
    #  dtype, so this one term is float32 arithmetic in the reference)
    log_scale = torch.log(action_scale.to(torch.float32) + 1e-6).to(mean.dtype)
    logp = logp - torch.log((1.0 - t * t) + 1e-6) - log_scale
    return t * action_scale, logp.sum(dim=1)


def critic_logits(qflat, obs_dim, act_dim, nr_atoms, obs, act):
    return forward(qflat, obs_dim + act_dim, CRITIC_HIDDEN, nr_atoms, torch.cat([obs, act], dim=1))


def adamw(p, g, m, v, step, lr, weight_decay, b1, b2, eps=1e-8):
    """One torch.optim.AdamW step (numpy arrays, `step` 1-based).  -> (p, m, v)."""
    p = p * (1.0 - lr * weight_decay)
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * g * g
    bc1, bc2 = 1.0 - b1 ** step, 1.0 - b2 ** step
    p = p - (lr / bc1) * m / (np.sqrt(v) / math.sqrt(bc2) + eps)
    return p, m, v


def clip_grad_norm(g, max_norm):
    """torch.nn.utils.clip_grad_norm_ on one flat gradient vector (fastsac.py:129-130, :218-219): g * min(1, c / (norm + 1e-6));
    returns (clipped gradient, the UN-clipped norm -- what the reference logs).  max_norm == -1: no clipping (fastsac.py:131-136)."""
    norm = float(np.linalg.norm(g))
    if max_norm == -1.0:
        return g, norm
    return g * min(1.0, max_norm / (norm + 1e-6)), norm


def polyak(target, params, tau):
    return (1.0 - tau) * target + tau * params


def critic_step(pflat, q1, q2, t1, t2, log_alpha, obs_dim, act_dim, nr_atoms, batch, noise_next, action_scale, hp, clipped):
    """batch = (states, next_states, actions, rewards, dones, truncations, n_steps), numpy float64.
    -> dict(q_loss, q_min, q_max, entropy, entropy_loss, g_q1, g_q2, g_log_alpha, next_log_probs)."""
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    s, s2, a, rew, done, trunc, nst = (t(x) for x in batch)
    scale = t(action_scale)
    with torch.no_grad():
        mean, ls = policy_forward(t(pflat), obs_dim, act_dim, s2, hp["log_std_min"], hp["log_std_max"])
        a2, lp2 = sample(mean, ls, t(noise_next), scale)
        nl1 = critic_logits(t(t1), obs_dim, act_dim, nr_atoms, s2, a2).numpy()
        nl2 = critic_logits(t(t2), obs_dim, act_dim, nr_atoms, s2, a2).numpy()
    alpha = math.exp(log_alpha)
    args = (rew.numpy(), done.numpy(), trunc.numpy(), nst.numpy(), lp2.numpy(), alpha, hp["gamma"], hp["v_min"], hp["v_max"])
    p1, v1 = c51.project(nl1, *args)
    p2, v2 = c51.project(nl2, *args)
    if clipped:
        sel = np.where((v1 < v2)[:, None], p1, p2)
        tgt1 = tgt2 = sel
    else:
        tgt1, tgt2 = p1, p2
    Q1, Q2 = t(q1).requires_grad_(True), t(q2).requires_grad_(True)
    l1 = critic_logits(Q1, obs_dim, act_dim, nr_atoms, s, a)
    l2 = critic_logits(Q2, obs_dim, act_dim, nr_atoms, s, a)
    loss = -(t(tgt1) * torch.log_softmax(l1, dim=1)).sum(dim=1).mean() - (t(tgt2) * torch.log_softmax(l2, dim=1)).sum(dim=1).mean()
    loss.backward()
    entropy = -lp2.numpy()
    return dict(q_loss=float(loss.detach()), q_min=float(v1.min()), q_max=float(v1.max()), entropy=float(entropy.mean()),
                entropy_loss=float((alpha * (entropy - hp["target_entropy"])).mean()), g_q1=Q1.grad.numpy(), g_q2=Q2.grad.numpy(),
                g_log_alpha=float(alpha * (entropy - hp["target_entropy"]).mean()), next_log_probs=lp2.numpy(),
                q1_logits=l1.detach().numpy(), q2_logits=l2.detach().numpy())


def policy_step(pflat, q1, q2, log_alpha, obs_dim, act_dim, nr_atoms, states, noise_cur, action_scale, hp, clipped):
    """-> dict(policy_loss, g_policy, log_probs, q_value)."""
    t = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    s, scale = t(states), t(action_scale)
    P = t(pflat).requires_grad_(True)
    mean, ls = policy_forward(P, obs_dim, act_dim, s, hp["log_std_min"], hp["log_std_max"])
    a, lp = sample(mean, ls, t(noise_cur), scale)
    z = torch.linspace(hp["v_min"], hp["v_max"], nr_atoms, dtype=torch.float64)
    v1 = (torch.softmax(critic_logits(t(q1), obs_dim, act_dim, nr_atoms, s, a), dim=1) * z).sum(dim=1)
    v2 = (torch.softmax(critic_logits(t(q2), obs_dim, act_dim, nr_atoms, s, a), dim=1) * z).sum(dim=1)
    q = torch.minimum(v1, v2) if clipped else (v1 + v2) / 2.0
    loss = (math.exp(log_alpha) * lp - q).mean()
    loss.backward()
    return dict(policy_loss=float(loss.detach()), g_policy=P.grad.numpy(), log_probs=lp.detach().numpy(), q_value=q.detach().numpy(),
                actions=a.detach().numpy())


def nstep_sample(ring, pos, size, capacity, n_steps, gamma, idx_t, idx_e):
    """ReplayBuffer.sample (replay_buffer.py:34-96) for GIVEN start rows idx_t and env columns idx_e (the reference draws them with
    torch.randint).  ring = dict(states, next_states, actions, rewards, dones, truncations) of [capacity, nr_envs, ...] numpy arrays.
    -> (states, next_states, actions, rewards, dones, truncations, effective_n_steps)."""
    B = idx_t.shape[0]
    if n_steps == 1:
        g = lambda k: ring[k][idx_t, idx_e]
        return g("states"), g("next_states"), g("actions"), g("rewards"), g("dones"), g("truncations"), np.ones(B, ring["dones"].dtype)
    trunc = ring["truncations"]
    if size >= capacity:
        trunc = trunc.copy()
        last = (pos - 1) % capacity
        trunc[last] = np.where(ring["dones"][last] > 0.0, trunc[last], 1.0)
    all_t = (idx_t[:, None] + np.arange(n_steps)[None, :]) % capacity
    env = np.broadcast_to(idx_e[:, None], all_t.shape)
    rew, dn, tr = ring["rewards"][all_t, env], ring["dones"][all_t, env], trunc[all_t, env]
    shifted = np.concatenate([np.zeros((B, 1), dn.dtype), dn[:, :-1]], axis=1)
    masks = np.cumprod(1.0 - shifted, axis=1)
    eff = masks.sum(axis=1)
    disc = (np.asarray(gamma, dtype=np.float32) ** np.arange(n_steps, dtype=np.float32)).astype(rew.dtype)
    rewards = (rew * masks * disc[None, :]).sum(axis=1)
    first = lambda x: np.where((x > 0.0).sum(axis=1) == 0, n_steps - 1, np.argmax(x > 0.0, axis=1))
    final = np.minimum(first(dn), first(tr))
    ft = all_t[np.arange(B), final]
    return (ring["states"][idx_t, idx_e], ring["next_states"][ft, idx_e], ring["actions"][idx_t, idx_e], rewards, ring["dones"][ft, idx_e],
            trunc[ft, idx_e], eff)
