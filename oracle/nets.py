"""Oracle (test infrastructure): the reference's MLP actor/critic restated in numpy.

Follows (math only; flax.linen semantics restated from its published behaviour,
flax<=0.12.0 is not under /root/reference):
  arch "A"  rl_x/algorithms/ppo/flax/policy.py:31-40, critic.py:22-30
            Dense(H)-tanh-Dense(H)-tanh-Dense(out)
  arch "B"  rl_x/algorithms/ppo/flax_full_jit/policy.py:30-42, critic.py:21-32
            Dense(512)-LayerNorm-ELU-Dense(256)-ELU-Dense(128)-ELU-Dense(out)
  Dense:     y = x @ kernel[in,out] + bias
  LayerNorm: eps=1e-6, var = max(0, E[x^2]-E[x]^2), y = (x-mu)*rsqrt(var+eps)*scale+bias
  elu:       x>0 ? x : expm1(x)

FLAT PARAMETER LAYOUT (shared with the HIP library, include/rlx_hip.h):
  for each hidden layer l: W_l[in,out] row-major, b_l[out], then (layer 0 of
  arch B only) ln_scale[out], ln_bias[out]; then head W[in,out], b[out]; the
  policy additionally ends with logstd[act_dim].
Arch "A" (Dense + tanh) and the SAC relu nets: PINNED against the reference's PyTorch modules executed in the authoring
container (tests/test_oracle_reference_pin.py).  Arch "B" (LayerNorm + ELU exists only in the JAX flavour): PARITY
UNPINNED by the reference; forward and manual backward pinned against float64 torch (torch.nn.functional.layer_norm / elu,
autograd; torch.nn.Linear / LayerNorm / ELU modules) in tests/test_oracle_ppo.py.
"""
import numpy as np

ACT_TANH, ACT_ELU, ACT_RELU = 0, 1, 2
LN_EPS = 1e-6


class MLPSpec:
    def __init__(self, in_dim, hidden, out_dim, act, ln_first, has_logstd):
        self.in_dim = int(in_dim)
        self.hidden = [int(h) for h in hidden]
        self.out_dim = int(out_dim)
        self.act = act
        self.ln_first = bool(ln_first)
        self.has_logstd = bool(has_logstd)
        # offsets
        off = 0
        self.layers = []  # dicts: W, b, (g, be), in, out
        d = self.in_dim
        for li, h in enumerate(self.hidden):
            L = {"in": d, "out": h, "W": off}
            off += d * h
            L["b"] = off
            off += h
            if self.ln_first and li == 0:
                L["g"] = off
                off += h
                L["be"] = off
                off += h
            self.layers.append(L)
            d = h
        self.head = {"in": d, "out": self.out_dim, "W": off}
        off += d * self.out_dim
        self.head["b"] = off
        off += self.out_dim
        if self.has_logstd:
            self.logstd = off
            off += self.out_dim
        self.n_params = off


def make_spec(arch, obs_dim, out_dim, is_policy, nr_hidden_units=256):
    if arch == "A":
        return MLPSpec(obs_dim, [nr_hidden_units, nr_hidden_units], out_dim, ACT_TANH, False, is_policy)
    if arch == "B":
        return MLPSpec(obs_dim, [512, 256, 128], out_dim, ACT_ELU, True, is_policy)
    raise ValueError(arch)


def orthogonal(rng, shape, scale):
    """Distribution-matched flax orthogonal init (QR of a normal matrix, sign fix)."""
    n_rows, n_cols = shape
    big, small = max(n_rows, n_cols), min(n_rows, n_cols)
    a = rng.standard_normal((big, small))
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if n_rows < n_cols:
        q = q.T
    return (scale * q).astype(np.float64)


def init_params(spec, rng, head_scale, std_dev=1.0, dtype=np.float32):
    """PPO init: orthogonal(sqrt2) trunk, orthogonal(head_scale) head, zero bias,
    LN scale 1 / bias 0, logstd = log(std_dev)  (ppo/flax_full_jit/policy.py:30-41)."""
    p = np.zeros(spec.n_params, dtype=np.float64)
    for L in spec.layers:
        p[L["W"]:L["W"] + L["in"] * L["out"]] = orthogonal(rng, (L["in"], L["out"]), np.sqrt(2)).ravel()
        if "g" in L:
            p[L["g"]:L["g"] + L["out"]] = 1.0
    H = spec.head
    p[H["W"]:H["W"] + H["in"] * H["out"]] = orthogonal(rng, (H["in"], H["out"]), head_scale).ravel()
    if spec.has_logstd:
        p[spec.logstd:spec.logstd + spec.out_dim] = np.log(std_dev)
    return p.astype(dtype)


def _act(z, act):
    if act == ACT_TANH:
        return np.tanh(z)
    if act == ACT_ELU:
        return np.where(z > 0, z, np.expm1(np.minimum(z, 0)))
    return np.maximum(z, 0)


def _act_grad_from_out(h, act):
    """d act / dz expressed with the OUTPUT h (what the HIP backward uses)."""
    if act == ACT_TANH:
        return 1 - h * h
    if act == ACT_ELU:
        return np.where(h > 0, 1.0, h + 1.0).astype(h.dtype)
    return (h > 0).astype(h.dtype)


def forward(spec, params, x):
    """Returns (out[n,out_dim], cache)."""
    dt = params.dtype
    x = x.astype(dt)
    cache = {"x": x, "h": [], "xhat": None, "rstd": None}
    h = x
    for li, L in enumerate(spec.layers):
        W = params[L["W"]:L["W"] + L["in"] * L["out"]].reshape(L["in"], L["out"])
        b = params[L["b"]:L["b"] + L["out"]]
        z = h @ W + b
        if "g" in L:
            mu = z.mean(axis=1, keepdims=True)
            var = np.maximum(0, (z * z).mean(axis=1, keepdims=True) - mu * mu)
            rstd = 1.0 / np.sqrt(var + dt.type(LN_EPS))
            xhat = (z - mu) * rstd
            cache["xhat"], cache["rstd"] = xhat, rstd
            z = xhat * params[L["g"]:L["g"] + L["out"]] + params[L["be"]:L["be"] + L["out"]]
        h = _act(z, spec.act).astype(dt)
        cache["h"].append(h)
    H = spec.head
    W = params[H["W"]:H["W"] + H["in"] * H["out"]].reshape(H["in"], H["out"])
    out = h @ W + params[H["b"]:H["b"] + H["out"]]
    return out.astype(dt), cache


def backward(spec, params, cache, d_out, need_dx=False):
    """Manual reverse pass.  d_out[n,out_dim] = dLoss/d(out).  Returns flat grads
    (logstd slot left 0 -- the loss fills it), and dLoss/dx if need_dx."""
    dt = params.dtype
    g = np.zeros(spec.n_params, dtype=dt)
    H = spec.head
    hs = cache["h"]
    h_last = hs[-1]
    W = params[H["W"]:H["W"] + H["in"] * H["out"]].reshape(H["in"], H["out"])
    g[H["W"]:H["W"] + H["in"] * H["out"]] = (h_last.T @ d_out).ravel()
    g[H["b"]:H["b"] + H["out"]] = d_out.sum(axis=0)
    dh = d_out @ W.T
    dx = None
    for li in range(len(spec.layers) - 1, -1, -1):
        L = spec.layers[li]
        dz = dh * _act_grad_from_out(hs[li], spec.act)
        if "g" in L:
            xhat, rstd = cache["xhat"], cache["rstd"]
            gam = params[L["g"]:L["g"] + L["out"]]
            g[L["g"]:L["g"] + L["out"]] = (dz * xhat).sum(axis=0)
            g[L["be"]:L["be"] + L["out"]] = dz.sum(axis=0)
            dxh = dz * gam
            m1 = dxh.mean(axis=1, keepdims=True)
            m2 = (dxh * xhat).mean(axis=1, keepdims=True)
            dz = rstd * (dxh - m1 - xhat * m2)
        inp = cache["x"] if li == 0 else hs[li - 1]
        W = params[L["W"]:L["W"] + L["in"] * L["out"]].reshape(L["in"], L["out"])
        g[L["W"]:L["W"] + L["in"] * L["out"]] = (inp.T @ dz).ravel()
        g[L["b"]:L["b"] + L["out"]] = dz.sum(axis=0)
        if li > 0 or need_dx:
            dh = dz @ W.T
            if li == 0:
                dx = dh
    return (g, dx) if need_dx else g


def torch_forward(spec, params_t, x_t):
    """Same forward in torch (for autograd cross-checks and the CPU baseline)."""
    import torch
    h = x_t
    for li, L in enumerate(spec.layers):
        W = params_t[L["W"]:L["W"] + L["in"] * L["out"]].view(L["in"], L["out"])
        z = h @ W + params_t[L["b"]:L["b"] + L["out"]]
        if "g" in L:
            mu = z.mean(dim=1, keepdim=True)
            var = torch.clamp((z * z).mean(dim=1, keepdim=True) - mu * mu, min=0)
            z = (z - mu) * torch.rsqrt(var + LN_EPS) * params_t[L["g"]:L["g"] + L["out"]] + params_t[L["be"]:L["be"] + L["out"]]
        if spec.act == ACT_TANH:
            h = torch.tanh(z)
        elif spec.act == ACT_ELU:
            h = torch.nn.functional.elu(z)
        else:
            h = torch.relu(z)
    H = spec.head
    W = params_t[H["W"]:H["W"] + H["in"] * H["out"]].view(H["in"], H["out"])
    return h @ W + params_t[H["b"]:H["b"] + H["out"]]
