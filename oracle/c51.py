"""CPU restatement of FastSAC's distributional (categorical, "C51") critic step -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

Follows rl_x/algorithms/fastsac/pytorch/fastsac.py, closure critic_and_entropy_loss_fn:
    :147-151  bootstrap = 1 - dones * (1 - truncations); discount = gamma ** n_steps * bootstrap;
              entropy-adjusted reward = rewards - discount * alpha * next_log_probs
    :152-162  target_z = clamp(r + discount * z_j, v_min, v_max); b = (target_z - v_min) / delta_z; l = floor(b), u = ceil(b);
              where b is integral: l -= 1 if l > 0, else u += 1 (so that the two weights u - b, b - l always sum to one)
    :164-190  next distributions = softmax(target-network logits); projection by two index_add_ passes per network
              (all (u - b) p_j into bin l_j, then all (b - l) p_j into bin u_j -- in that order, j ascending)
    :192-201  q_k_next_value = sum(proj_k * z); clipped double Q: BOTH critics learn from the projection whose value is smaller
    :203-209  q_k_loss = -mean_b sum_j target_kj * log_softmax(q_k logits)_j; q_loss = q1_loss + q2_loss
    :212-213  q_min / q_max = extrema of q1_next_value

Pinned by tests/golden/reference_c51.npz: outputs of that closure itself (compiled from the reference file, Q networks replaced by
tables of logits so that the parameter gradients are d q_loss / d logits), float64 and float32."""
import numpy as np


def softmax(x):
    m = x.max(axis=1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=1, keepdims=True)


def log_softmax(x):
    m = x.max(axis=1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=1, keepdims=True))


def project(next_logits, rewards, dones, truncations, n_steps, next_log_probs, alpha, gamma, v_min, v_max):
    """-> (projected target distribution [B, NA], its expectation [B])."""
    dt = next_logits.dtype
    B, NA = next_logits.shape
    z = np.linspace(v_min, v_max, NA).astype(dt)
    delta_z = (v_max - v_min) / (NA - 1)
    bootstrap = 1.0 - dones * (1.0 - truncations)
    discount = (np.asarray(gamma, dt) ** n_steps * bootstrap).astype(dt)
    r = (rewards - discount * np.asarray(alpha, dt) * next_log_probs).astype(dt)
    tz = np.clip(r[:, None] + discount[:, None] * z[None, :], dt.type(v_min), dt.type(v_max))
    b = ((tz - dt.type(v_min)) / dt.type(delta_z)).astype(dt)
    l = np.floor(b).astype(np.int64)
    u = np.ceil(b).astype(np.int64)
    is_int = l == u
    l = np.where(is_int & (l > 0), l - 1, l)
    u = np.where(is_int & (l == 0) & (np.floor(b).astype(np.int64) == 0), u + 1, u)   # u_mask = is_int & (original l == 0)
    p = softmax(next_logits)
    wl = (u.astype(dt) - b) * p
    wu = (b - l.astype(dt)) * p
    proj = np.zeros((B, NA), dt)
    for j in range(NA):            # index_add_ order: every l contribution (j ascending), then every u contribution
        np.add.at(proj, (np.arange(B), l[:, j]), wl[:, j])
    for j in range(NA):
        np.add.at(proj, (np.arange(B), u[:, j]), wu[:, j])
    return proj, (proj * z[None, :]).sum(axis=1)


def critic_loss(q1_logits, q2_logits, q1_next_logits, q2_next_logits, rewards, dones, truncations, n_steps, next_log_probs, alpha,
                gamma, v_min, v_max, clipped_double_q):
    """-> dict(q_loss, q_min, q_max, d_q1, d_q2 (gradients of q_loss w.r.t. the logits), target1, target2)."""
    args = (rewards, dones, truncations, n_steps, next_log_probs, alpha, gamma, v_min, v_max)
    p1, v1 = project(q1_next_logits, *args)
    p2, v2 = project(q2_next_logits, *args)
    if clipped_double_q:
        t1 = t2 = np.where((v1 < v2)[:, None], p1, p2)
    else:
        t1, t2 = p1, p2
    B = q1_logits.shape[0]
    out = {"q_min": v1.min(), "q_max": v1.max(), "target1": t1, "target2": t2}
    loss = 0.0
    for name, logits, t in (("d_q1", q1_logits, t1), ("d_q2", q2_logits, t2)):
        loss = loss + (-(t * log_softmax(logits)).sum(axis=1)).mean()
        out[name] = (softmax(logits) * t.sum(axis=1, keepdims=True) - t) / B        # d(-sum_j t_j log_softmax_j) / d logits
    out["q_loss"] = loss
    return out
