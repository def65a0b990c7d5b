"""Oracle (test infrastructure): CPU restatement of the build's synthetic
random-observation environment (the reference has NO
synthetic env; this is the build's own definition, shaped like the reference's
vectorised TORCH/JAX-interface envs, e.g.
rl_x/environments/custom_mujoco/ant/warp_torch/environment.py:142-186 for the
auto-reset + final-observation contract and
rl_x/environments/custom_mujoco/ant/mjx/state.py:7-17 for `actual_next_observation`).

Definition (all draws from a counter-based RNG so results do not depend on the
GPU count):  W(n, t, j) = threefry2x32(key=(seed, n_global), ctr=(t, j)) -> 2 words
  next_obs[d]   = N(W(n, t, d//2)[d%2])
  reward        = -(1/A) sum_j (clip(a_j,-1,1) - tanh(obs[j % O]))^2 + noise * N(W(n,t,64)[0])
  terminated    = U01(W(n,t,64)[1]) < p_term
  truncated     = episode_step >= horizon          (per-env start phase (n*7919) % horizon)
  on done:      policy continues from reset_obs[d] = N(W(n, t, 128 + d//2)[d%2]);
                the rollout buffer's next_state keeps the final observation.
"""
import numpy as np
from . import prng

U32 = np.uint32
RESET_T = 0xFFFFFFFF
STREAM_MISC = 64
STREAM_RESET = 128
PHASE_MULT = 7919


def _normal_from_bits(bits):
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    f = prng._bits_to_unit_float(bits)
    u = np.maximum(lo, (f * (np.float32(1.0) - lo) + lo).astype(np.float32))
    return (np.float32(np.sqrt(2)) * prng.erfinv_f32(u)).astype(np.float32)


def _obs_draw(seed, n_global, t, obs_dim, stream0):
    d = np.arange(obs_dim)
    w0, w1 = prng.threefry2x32(U32(seed), n_global[:, None].astype(U32), U32(t), (stream0 + d // 2)[None, :].astype(U32))
    bits = np.where((d % 2)[None, :] == 0, w0, w1)
    return _normal_from_bits(bits)


class RandomObsEnvOracle:
    def __init__(self, seed, nr_envs, obs_dim, act_dim, horizon=1000, p_term=1e-3, reward_noise=0.1,
                 env_id_offset=0):
        self.seed, self.N, self.O, self.A = int(seed), nr_envs, obs_dim, act_dim
        self.horizon, self.p_term, self.noise = horizon, np.float32(p_term), np.float32(reward_noise)
        self.ids = (np.arange(nr_envs) + env_id_offset).astype(np.uint32)
        self.t = 0

    def reset(self):
        self.t = 0
        self.obs = _obs_draw(self.seed, self.ids, RESET_T, self.O, STREAM_RESET)
        self.ep_step = ((self.ids.astype(np.int64) * PHASE_MULT) % self.horizon).astype(np.int32)
        self.ep_ret = np.zeros(self.N, np.float32)
        self.last_ret = np.zeros(self.N, np.float32)
        self.last_len = np.zeros(self.N, np.float32)
        return self.obs.copy()

    def step(self, action):
        a = np.clip(action.astype(np.float32), -1, 1)
        j = np.arange(self.A)
        target = np.tanh(self.obs[:, j % self.O]).astype(np.float32)
        w0, w1 = prng.threefry2x32(U32(self.seed), self.ids, U32(self.t), U32(STREAM_MISC))
        zr = _normal_from_bits(w0)
        ut = prng._bits_to_unit_float(w1)
        diff = a - target
        reward = (-(diff * diff).sum(axis=1, dtype=np.float32) / np.float32(self.A) + self.noise * zr).astype(np.float32)
        nobs = _obs_draw(self.seed, self.ids, self.t, self.O, 0)
        robs = _obs_draw(self.seed, self.ids, self.t, self.O, STREAM_RESET)
        terminated = ut < self.p_term
        self.ep_step = self.ep_step + 1
        truncated = self.ep_step >= self.horizon
        done = terminated | truncated
        self.ep_ret = (self.ep_ret + reward).astype(np.float32)
        self.last_ret = np.where(done, self.ep_ret, self.last_ret)
        self.last_len = np.where(done, self.ep_step.astype(np.float32), self.last_len)
        self.ep_ret = np.where(done, np.float32(0), self.ep_ret)
        self.ep_step = np.where(done, 0, self.ep_step).astype(np.int32)
        self.obs = np.where(done[:, None], robs, nobs)
        self.t += 1
        return self.obs.copy(), nobs, reward, terminated, truncated, done
