"""synthetic.random_obs -- the benchmark env of BASELINE.json configs[1..2]: a vectorised,
device-resident, auto-resetting TORCH-interface environment whose transition is one HIP kernel
(`rlx_env_step_f32`, rl-x_amd/csrc/env.hip).  The reference has no synthetic env; the object
contract is the one its PPO loops use:
  reset() -> (obs, info); step(action) -> (next_obs, reward, terminated, truncated, info);
  close(); single_observation_space / single_action_space; general_properties;
  get_logging_info_dict / get_final_observation_at_index / get_final_info_value_at_index
  (cf. rl_x/environments/custom_mujoco/ant/warp_torch/{environment.py:142-186,wrappers.py:4-51}).
Extra, used by `ppo.hip`'s fast path: `step_into(action, final_obs_out, reward_out,
terminated_out)` writes the transition straight into the rollout-buffer rows.
"""
import numpy as np


class Box:
    """Minimal stand-in for gymnasium.spaces.Box (gymnasium is not a dependency)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = dtype
        self.low = np.full(self.shape, low, dtype=dtype)
        self.high = np.full(self.shape, high, dtype=dtype)
        self._rng = np.random.default_rng(0)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return self._rng.uniform(lo, hi).astype(self.dtype)


def _dist_info():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


class RandomObsEnv:
    step_into_prev_obs = True     # step_into(..., prev_obs_out=rows) is supported
    def __init__(self, env_config, eval_stream=False):
        import torch
        from rlx_amd.hip import Ctx
        self.torch = torch
        self.rank, self.world = _dist_info()
        self.nr_envs_global = int(env_config.nr_envs)
        if self.nr_envs_global % self.world != 0:
            raise ValueError("environment.nr_envs must be divisible by the number of ranks")
        self.nr_envs = self.nr_envs_global // self.world          # local shard
        self.env_id_offset = self.rank * self.nr_envs
        self.seed = (int(env_config.seed) + (0x9E3779B9 if eval_stream else 0)) & 0xFFFFFFFF
        self.obs_dim = int(env_config.obs_dim)
        self.act_dim = int(env_config.act_dim)
        self.horizon = int(env_config.horizon)
        self.p_term = float(env_config.termination_probability)
        self.reward_noise = float(env_config.reward_noise)
        if env_config.device != "gpu":
            raise ValueError("synthetic.random_obs is device-resident: --environment.device must be 'gpu'")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.ctx = Ctx(self.device.index)
        self.single_observation_space = Box(-np.inf, np.inf, (self.obs_dim,))
        self.single_action_space = Box(-1.0, 1.0, (self.act_dim,))
        N, O = self.nr_envs, self.obs_dim
        f = dict(device=self.device, dtype=torch.float32)
        self.obs = torch.zeros(N, O, **f)
        self.ep_step = torch.zeros(N, device=self.device, dtype=torch.int32)
        self.ep_ret, self.last_ret, self.last_len = (torch.zeros(N, **f) for _ in range(3))
        self.episode_stats = torch.zeros(4, **f)   # finished episodes, sum of returns, sum of lengths
        self._fin, self._rew, self._term, self._trunc = (None,) * 4
        self.t = 0

    # ------------------------------------------------------------------ gym-like API
    def reset(self):
        self.t = 0
        self.episode_stats.zero_()
        self.ctx.env_reset(self.seed, self.env_id_offset, self.horizon, self.obs, self.ep_step, self.ep_ret,
                           self.last_ret, self.last_len)
        return self.obs, {}

    def step_into(self, action, final_obs_out, reward_out, terminated_out, truncated_out=None, prev_obs_out=None):
        """One transition; outputs land in caller-provided (rollout-buffer) rows.  `self.obs`
        is advanced in place to the post-reset next observation; prev_obs_out (optional) receives the observation
        it held BEFORE the step (the replay ring's `states` row of this transition)."""
        if truncated_out is None:
            if self._trunc is None:
                self._trunc = self.torch.empty(self.nr_envs, device=self.device)
            truncated_out = self._trunc
        self.ctx.env_step(self.seed, self.env_id_offset, self.t, self.horizon, self.p_term, self.reward_noise, action,
                          self.obs, final_obs_out, reward_out, terminated_out, truncated_out, self.ep_step,
                          self.ep_ret, self.last_ret, self.last_len, self.episode_stats, prev_obs_out=prev_obs_out)
        self.t += 1

    def fused_args(self, final_obs_out, reward_out, terminated_out):
        """State handed to `rlx_ppo_rollout_step_f32` so the env transition runs inside the acting kernel
        (the caller advances `self.t` with `fused_advance`)."""
        return dict(seed=self.seed, env_id_offset=self.env_id_offset, t=self.t, horizon=self.horizon,
                    p_term=self.p_term, reward_noise=self.reward_noise, final_obs=final_obs_out, reward=reward_out,
                    terminated=terminated_out, ep_step=self.ep_step, ep_ret=self.ep_ret, last_ret=self.last_ret,
                    last_len=self.last_len, episode_stats=self.episode_stats)

    def fused_advance(self, steps=1):
        self.t += steps

    def step(self, action):
        t = self.torch
        if self._fin is None:
            self._fin = t.empty(self.nr_envs, self.obs_dim, device=self.device)
            self._rew, self._term, self._trunc = (t.empty(self.nr_envs, device=self.device) for _ in range(3))
        self.step_into(action.contiguous(), self._fin, self._rew, self._term, self._trunc)
        terminated, truncated = self._term > 0.5, self._trunc > 0.5
        info = {"dones": terminated | truncated, "final_observation": self._fin,
                "rollout/episode_return": self.last_ret, "rollout/episode_length": self.last_len}
        return self.obs, self._rew, terminated, truncated, info

    def close(self):
        pass

    # ------------------------------------------------------------------ evaluation on the training env
    def snapshot(self):
        """Everything `reset` / `step` mutate; `restore` puts it back IN PLACE (callers hold `self.obs`)."""
        return (self.t, self.obs.clone(), self.ep_step.clone(), self.ep_ret.clone(), self.last_ret.clone(),
                self.last_len.clone(), self.episode_stats.clone())

    def restore(self, snap):
        self.t = snap[0]
        for dst, src in zip((self.obs, self.ep_step, self.ep_ret, self.last_ret, self.last_len, self.episode_stats), snap[1:]):
            dst.copy_(src)

    # ------------------------------------------------------------------ RLXInfo helpers
    def get_logging_info_dict(self, info):
        out = {}
        done_mask = info.get("dones")
        for key, value in info.items():
            if key in ("dones", "final_observation") or not self.torch.is_tensor(value):
                continue
            v = value
            if key.startswith("rollout/") and done_mask is not None:
                if not bool(done_mask.any()):
                    continue
                v = v[done_mask]
            out[key] = v.detach().cpu().tolist()
        return out

    def get_final_observation_at_index(self, info, index):
        return info["final_observation"][index]

    def get_final_info_value_at_index(self, info, key, index):
        value = info.get(key, info.get(f"rollout/{key}"))
        if value is None:
            raise KeyError(f"info has no key {key!r}")
        item = value[index]
        return item.item() if self.torch.is_tensor(item) else item

    def pop_episode_stats(self):
        """(finished episodes, mean return, mean length) since the last call -- ONE D2H copy,
        replacing the reference's per-step `.cpu().tolist()` (warp_torch/wrappers.py:15-30)."""
        s = self.episode_stats.cpu().tolist()
        self.episode_stats.zero_()
        n = s[0]
        return int(n), (s[1] / n if n else float("nan")), (s[2] / n if n else float("nan"))
