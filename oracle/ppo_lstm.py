"""Oracle (test infrastructure): the reference PPO+LSTM policy and loss restated in torch-CPU.

Follows:
  Policy (full-jit arch)     rl_x/algorithms/ppo_lstm/flax_full_jit/policy.py:32-142
     lstm_obs_encode :88-93, obs_encode :80-85, decode :96-118 ("concat" and "film" methods), apply_one_step :121-131,
     forward_sequence :134-142 (carry multiplied by (1 - done[t-1]) BEFORE consuming obs[t])
  rollout carry masking      rl_x/algorithms/ppo_lstm/flax_full_jit/ppo_lstm.py:148-149 (after the env step)
  sequence minibatches       ppo_lstm.py:226-259 (env-index permutation [E, N] -> [E*M, minibatch_size // nr_steps])
  loss_fn                    ppo_lstm.py:181-216 (same PPO loss, meaned over time and envs)
flax.linen.OptimizedLSTMCell (third party, flax<=0.12.0) restated: i = sig(x Wii + h Whi + bhi), f, g = tanh(.), o;
c' = f c + i g; h' = o tanh(c'); carry order (c, h).
PPO+GRU (rl_x/algorithms/ppo_gru/flax_full_jit/policy.py: the same policy with nn.GRUCell, single carry h) is the
cell="gru" variant: flax.linen.GRUCell restated: r = sig(x Wir + bir + h Whr), z = sig(x Wiz + biz + h Whz),
n = tanh(x Win + bin + r * (h Whn + bhn)), h' = (1 - z) n + z h.
Gradients come from torch.autograd in float64 (the independent check of the HIP BPTT kernels).
PARITY UNPINNED by the reference (no tests, JAX not installable).

FLAT PARAMETER LAYOUT of the recurrent policy (shared with librlxhip.so, include/rlx_hip.h):
  enc_l: W[O,E] b[E] ln_g[E] ln_b[E] | enc_o: same (absent when share_encoder)
  lstm : Wi[E,4H] (gate blocks i,f,g,o; no bias)  Wh[H,4H]  bh[4H] | lstm_ln: g[H] b[H]
  (cell="gru": gru: Wi[E,3H] (gate blocks r,z,n)  bi[3H]  Wh_rz[H,2H]  Wh_n[H,H]  bhn[H] | lstm_ln as above)
  (combine="film": film: W[H,2E] (columns [0,E) gamma kernel, [E,2E) beta kernel) b[2E]; torso1 W is then [E,D1])
  torso1: W[E+H,D1] b[D1] ln_g[D1] ln_b[D1] | torso2: W[D1,D2] b[D2] | torso3: W[D2,D3] b[D3]
  head : W[D3,A] b[A] | logstd[A]
"""
import math

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)
LN_EPS = 1e-6


class LstmPolicySpec:
    def __init__(self, obs_dim, act_dim, enc_dim=128, lstm_hidden=64, torso=(512, 256, 128), share_encoder=False, cell="lstm",
                 combine="concat"):
        assert cell in ("lstm", "gru") and combine in ("concat", "film")
        self.cell = cell
        self.combine = combine
        self.O, self.A, self.E, self.H = obs_dim, act_dim, enc_dim, lstm_hidden
        self.torso = tuple(torso)
        self.share = bool(share_encoder)
        off = 0
        self.off = {}

        def take(name, n):
            nonlocal off
            self.off[name] = (off, n)
            off += n
        O, A, E, H = self.O, self.A, self.E, self.H
        D1, D2, D3 = self.torso
        for enc in (["enc_l"] if self.share else ["enc_l", "enc_o"]):
            take(enc + ".W", O * E); take(enc + ".b", E); take(enc + ".g", E); take(enc + ".be", E)
        if cell == "lstm":
            take("lstm.Wi", E * 4 * H); take("lstm.Wh", H * 4 * H); take("lstm.bh", 4 * H)
        else:
            take("gru.Wi", E * 3 * H); take("gru.bi", 3 * H); take("gru.Wh_rz", H * 2 * H); take("gru.Wh_n", H * H)
            take("gru.bhn", H)
        take("lstm_ln.g", H); take("lstm_ln.be", H)
        if combine == "film":
            take("film.W", H * 2 * E); take("film.b", 2 * E)
        self.K1 = E if combine == "film" else E + H
        take("t1.W", self.K1 * D1); take("t1.b", D1); take("t1.g", D1); take("t1.be", D1)
        take("t2.W", D1 * D2); take("t2.b", D2)
        take("t3.W", D2 * D3); take("t3.b", D3)
        take("head.W", D3 * A); take("head.b", A)
        take("logstd", A)
        self.n_params = off

    def get(self, p, name, shape=None):
        o, n = self.off[name]
        v = p[o:o + n]
        return v.reshape(shape) if shape else v


def init_params(spec, rng, std_dev=1.0):
    """orthogonal(sqrt 2) Dense kernels, orthogonal(0.01) mean head, LN scale 1, zero biases; LSTM: lecun-normal input
    kernels, orthogonal recurrent kernels (flax defaults); logstd = log(std_dev).  Distribution-matched only."""
    from .nets import orthogonal
    p = np.zeros(spec.n_params, dtype=np.float64)
    O, A, E, H = spec.O, spec.A, spec.E, spec.H
    D1, D2, D3 = spec.torso

    def put(name, arr):
        o, n = spec.off[name]
        p[o:o + n] = np.asarray(arr).ravel()
    for enc in (["enc_l"] if spec.share else ["enc_l", "enc_o"]):
        put(enc + ".W", orthogonal(rng, (O, E), math.sqrt(2)))
        put(enc + ".g", np.ones(E))
    if spec.cell == "lstm":
        put("lstm.Wi", rng.standard_normal((E, 4 * H)) / math.sqrt(E))
        put("lstm.Wh", np.concatenate([orthogonal(rng, (H, H), 1.0) for _ in range(4)], axis=1))
    else:
        put("gru.Wi", rng.standard_normal((E, 3 * H)) / math.sqrt(E))
        put("gru.Wh_rz", np.concatenate([orthogonal(rng, (H, H), 1.0) for _ in range(2)], axis=1))
        put("gru.Wh_n", orthogonal(rng, (H, H), 1.0))
    put("lstm_ln.g", np.ones(H))
    if spec.combine == "film":      # lstm_film_gamma / lstm_film_beta: two Dense(E) with orthogonal(sqrt 2) kernels, side by side
        put("film.W", np.concatenate([orthogonal(rng, (H, E), math.sqrt(2)) for _ in range(2)], axis=1))
    put("t1.W", orthogonal(rng, (spec.K1, D1), math.sqrt(2))); put("t1.g", np.ones(D1))
    put("t2.W", orthogonal(rng, (D1, D2), math.sqrt(2)))
    put("t3.W", orthogonal(rng, (D2, D3), math.sqrt(2)))
    put("head.W", orthogonal(rng, (D3, A), 0.01))
    put("logstd", np.full(A, math.log(std_dev)))
    return p


def _ln(x, g, b):
    import torch
    mu = x.mean(dim=-1, keepdim=True)
    var = torch.clamp((x * x).mean(dim=-1, keepdim=True) - mu * mu, min=0)
    return (x - mu) * torch.rsqrt(var + LN_EPS) * g + b


def _encode(spec, p, obs, which):
    import torch.nn.functional as F
    E = spec.E
    z = obs @ spec.get(p, which + ".W", (spec.O, E)) + spec.get(p, which + ".b")
    return F.elu(_ln(z, spec.get(p, which + ".g"), spec.get(p, which + ".be")))


def lstm_cell(spec, p, c, h, x):
    import torch
    H = spec.H
    gates = x @ spec.get(p, "lstm.Wi", (spec.E, 4 * H)) + h @ spec.get(p, "lstm.Wh", (H, 4 * H)) + spec.get(p, "lstm.bh")
    i, f, g, o = gates[..., :H], gates[..., H:2 * H], gates[..., 2 * H:3 * H], gates[..., 3 * H:]
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return c2, h2


def gru_cell(spec, p, h, x):
    """flax.linen.GRUCell; carry = h."""
    import torch
    H, E = spec.H, spec.E
    gx = x @ spec.get(p, "gru.Wi", (E, 3 * H)) + spec.get(p, "gru.bi")
    grz = h @ spec.get(p, "gru.Wh_rz", (H, 2 * H))
    hn = h @ spec.get(p, "gru.Wh_n", (H, H)) + spec.get(p, "gru.bhn")
    r = torch.sigmoid(gx[..., :H] + grz[..., :H])
    z = torch.sigmoid(gx[..., H:2 * H] + grz[..., H:])
    n = torch.tanh(gx[..., 2 * H:] + r * hn)
    return (1.0 - z) * n + z * h


def decode(spec, p, obs_latent, lstm_h):
    import torch
    import torch.nn.functional as F
    D1, D2, D3 = spec.torso
    lat = F.elu(_ln(lstm_h, spec.get(p, "lstm_ln.g"), spec.get(p, "lstm_ln.be")))
    if spec.combine == "film":      # policy.py:97-100: gamma = Dense(E)(lat); beta = Dense(E)(lat); x = obs_latent * gamma + beta
        gb = lat @ spec.get(p, "film.W", (spec.H, 2 * spec.E)) + spec.get(p, "film.b")
        x = obs_latent * gb[..., :spec.E] + gb[..., spec.E:]
    else:
        x = torch.cat([obs_latent, lat], dim=-1)
    h = F.elu(_ln(x @ spec.get(p, "t1.W", (spec.K1, D1)) + spec.get(p, "t1.b"), spec.get(p, "t1.g"), spec.get(p, "t1.be")))
    h = F.elu(h @ spec.get(p, "t2.W", (D1, D2)) + spec.get(p, "t2.b"))
    h = F.elu(h @ spec.get(p, "t3.W", (D2, D3)) + spec.get(p, "t3.b"))
    return h @ spec.get(p, "head.W", (D3, spec.A)) + spec.get(p, "head.b")


def apply_one_step(spec, p, obs, c, h):
    """policy.py:121-131.  obs [n,O], carry (c,h) [n,H] -> mean [n,A], new carry."""
    lat_l = _encode(spec, p, obs, "enc_l")
    if spec.cell == "gru":       # single carry: c is carried along untouched
        c2, h2 = c, gru_cell(spec, p, h, lat_l)
    else:
        c2, h2 = lstm_cell(spec, p, c, h, lat_l)
    lat_o = lat_l if spec.share else _encode(spec, p, obs, "enc_o")
    return decode(spec, p, lat_o, h2), c2, h2


def forward_sequence(spec, p, obs_seq, done_seq, c0, h0):
    """policy.py:134-142, batched over envs: obs_seq [T,n,O], done_seq [T,n], carry [n,H] -> mean [T,n,A]."""
    import torch
    T = obs_seq.shape[0]
    c, h = c0, h0
    means = []
    for t in range(T):
        if t > 0:
            m = (1.0 - done_seq[t - 1])[:, None]
            c, h = c * m, h * m
        mean, c, h = apply_one_step(spec, p, obs_seq[t], c, h)
        means.append(mean)
    return torch.stack(means)


def ppo_lstm_loss(spec, p, cspec, cp, obs_seq, act_seq, logp_seq, ret_seq, adv_seq, done_seq, c0, h0, clip, ent_c, v_c):
    """loss_fn (ppo_lstm.py:181-216) meaned over time and envs.  adv_seq is already normalised.
    Returns (loss, metrics dict) as torch scalars (differentiable w.r.t. p and cp)."""
    import torch
    from . import nets
    mean = forward_sequence(spec, p, obs_seq, done_seq, c0, h0)
    logstd = spec.get(p, "logstd")[None, None, :]
    std = torch.exp(logstd)
    nlp = (-0.5 * ((act_seq - mean) / std) ** 2 - 0.5 * LOG_2PI - logstd).sum(-1)
    entropy = (logstd + 0.5 * math.log(2.0 * math.pi * math.e)).sum()
    logratio = nlp - logp_seq
    ratio = torch.exp(logratio)
    pg = torch.maximum(-adv_seq * ratio, -adv_seq * torch.clamp(ratio, 1 - clip, 1 + clip))
    T, n, O = obs_seq.shape
    v = nets.torch_forward(cspec, cp, obs_seq.reshape(T * n, O)).reshape(T, n)
    vl = 0.5 * (v - ret_seq) ** 2
    loss = (pg - ent_c * entropy + v_c * vl).mean()
    metrics = {"loss/policy_gradient_loss": pg.mean(), "loss/critic_loss": vl.mean(), "loss/entropy_loss": entropy,
               "policy_ratio/approx_kl": ((ratio - 1) - logratio).mean(),
               "policy_ratio/clip_fraction": (torch.abs(ratio - 1) > clip).double().mean()}
    return loss, metrics


def env_minibatch_indices(key, nr_envs, nr_epochs, nr_minibatches, nr_minibatch_envs, partitionable=True):
    """ppo_lstm.py:226-229: key, sub = split(key); permutation of tile(arange(N), (E,1)) along axis 1."""
    from . import prng
    ks = prng.split(key, 2, partitionable)
    idx = prng.permutation_rows(ks[1], np.tile(np.arange(nr_envs, dtype=np.int32), (nr_epochs, 1)), partitionable)
    return ks[0], idx.reshape(nr_epochs * nr_minibatches, nr_minibatch_envs)
