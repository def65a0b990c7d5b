"""synthetic.numpy_obs -- a HOST-side (numpy) vectorised env with the object contract of the reference's Gymnasium
wrappers (rl_x/environments/gym/mujoco/humanoid_v4/wrappers.py: RLXInfo -> get_logging_info_dict,
get_final_observation_at_index, get_final_info_value_at_index; auto-reset with the final observation in `info`).
It exists to exercise the host-env ingestion path of `ppo.hip` (NUMPY data interface: actions D2H, transitions H2D
through pinned staging buffers) with the same task as synthetic.random_obs:
    reward = -mean_j (clip(a_j, -1, 1) - tanh(obs_j))^2 + noise,   obs' ~ N(0, 1),
    terminated with probability p, truncated at `horizon` steps."""
import numpy as np

from rlx_amd.environments.synthetic.random_obs.environment import Box


class NumpyObsEnv:
    def __init__(self, env_config, eval_stream=False):
        self.nr_envs = int(env_config.nr_envs)
        self.obs_dim = int(env_config.obs_dim)
        self.act_dim = int(env_config.act_dim)
        self.horizon = int(env_config.horizon)
        self.p_term = float(env_config.termination_probability)
        self.reward_noise = float(env_config.reward_noise)
        self.rng = np.random.default_rng([int(env_config.seed), 1 if eval_stream else 0])
        self.single_observation_space = Box(-np.inf, np.inf, (self.obs_dim,))
        self.single_action_space = Box(-1.0, 1.0, (self.act_dim,))
        self.obs = np.zeros((self.nr_envs, self.obs_dim), np.float32)
        self.ep_step = np.zeros(self.nr_envs, np.int64)
        self.ep_ret = np.zeros(self.nr_envs, np.float64)

    def reset(self):
        self.obs = self.rng.standard_normal((self.nr_envs, self.obs_dim)).astype(np.float32)
        self.ep_step[:] = self.rng.integers(0, self.horizon, self.nr_envs)   # staggered episode phases
        self.ep_ret[:] = 0.0
        return self.obs.copy(), {}

    def step(self, action):
        action = np.asarray(action, dtype=np.float32).reshape(self.nr_envs, self.act_dim)
        k = min(self.obs_dim, self.act_dim)
        err = np.clip(action[:, :k], -1.0, 1.0) - np.tanh(self.obs[:, :k])
        reward = (-np.mean(err * err, axis=1) + self.reward_noise * self.rng.standard_normal(self.nr_envs)).astype(np.float32)
        self.ep_step += 1
        self.ep_ret += reward
        terminated = self.rng.random(self.nr_envs) < self.p_term
        truncated = (self.ep_step >= self.horizon) & ~terminated
        done = terminated | truncated
        final_obs = self.rng.standard_normal((self.nr_envs, self.obs_dim)).astype(np.float32)   # s_{t+1} of every env
        next_obs = final_obs.copy()
        info = {"final_observation": final_obs, "done": done,
                "episode_return": self.ep_ret.copy(), "episode_length": self.ep_step.copy()}
        if done.any():
            n = int(done.sum())
            next_obs[done] = self.rng.standard_normal((n, self.obs_dim)).astype(np.float32)     # auto-reset
            self.ep_step[done] = 0
            self.ep_ret[done] = 0.0
        self.obs = next_obs
        return next_obs.copy(), reward, terminated, truncated, info

    def close(self):
        pass

    # ------------------------------------------------------------------ RLXInfo helpers
    def get_logging_info_dict(self, info):
        done = info.get("done")
        if done is None or not done.any():
            return {}
        return {"rollout/episode_return": info["episode_return"][done].tolist(),
                "rollout/episode_length": info["episode_length"][done].tolist()}

    def get_final_observation_at_index(self, info, index):
        return info["final_observation"][index]

    def get_final_info_value_at_index(self, info, key, index):
        return info[key][index]
