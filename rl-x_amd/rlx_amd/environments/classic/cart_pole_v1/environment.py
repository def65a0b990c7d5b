"""classic.cart_pole_v1 -- BASELINE.json configs[0]: CartPole-v1 as a HOST-side numpy vector env (gymnasium is not a
dependency of this build, so the classic-control task is restated from its published constants: gravity 9.8, cart mass
1.0, pole mass 0.1, half pole length 0.5, force 10 N, tau 0.02 s, explicit Euler; termination |x| > 2.4 or |theta| > 12
degrees; truncation after 500 steps; reward 1 per step; reset state uniform in [-0.05, 0.05]^4).

Object contract = the reference's wrapped Gymnasium vector env (rl_x/environments/gym/classic/cart_pole_v1/wrappers.py):
auto-reset with the final observation in `info`, RLXInfo helpers (get_logging_info_dict, get_final_observation_at_index,
get_final_info_value_at_index, get_single_action_logit_size), episode statistics as RecordEpisodeStatistics reports them.
DISCRETE action space: `step` takes one action index per env (any numeric dtype / shape [N] or [N, 1])."""
import numpy as np

from rlx_amd.environments.synthetic.random_obs.environment import Box

GRAVITY, MASS_CART, MASS_POLE, LENGTH, FORCE_MAG, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
TOTAL_MASS = MASS_CART + MASS_POLE
POLEMASS_LENGTH = MASS_POLE * LENGTH
X_LIMIT, THETA_LIMIT = 2.4, 12 * 2 * np.pi / 360
MAX_STEPS = 500


class Discrete:
    """Minimal stand-in for gymnasium.spaces.Discrete."""

    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64


class CartPoleVecEnv:
    def __init__(self, env_config, eval_stream=False):
        self.nr_envs = int(env_config.nr_envs)
        self.rng = np.random.default_rng([int(env_config.seed), 1 if eval_stream else 0])
        high = np.array([X_LIMIT * 2, np.inf, THETA_LIMIT * 2, np.inf], dtype=np.float32)
        self.single_observation_space = Box(-1.0, 1.0, (4,))
        self.single_observation_space.low, self.single_observation_space.high = -high, high
        self.single_action_space = Discrete(2)
        self.state = np.zeros((self.nr_envs, 4), np.float64)
        self.ep_step = np.zeros(self.nr_envs, np.int64)
        self.ep_ret = np.zeros(self.nr_envs, np.float64)

    def _reset_rows(self, rows):
        self.state[rows] = self.rng.uniform(-0.05, 0.05, size=(int(np.sum(rows)) if rows.dtype == bool else len(rows), 4))

    def reset(self):
        self._reset_rows(np.ones(self.nr_envs, bool))
        self.ep_step[:] = 0
        self.ep_ret[:] = 0.0
        return self.state.astype(np.float32), {}

    def step(self, action):
        a = np.asarray(action).reshape(self.nr_envs).astype(np.int64)
        x, x_dot, theta, theta_dot = self.state.T
        force = np.where(a == 1, FORCE_MAG, -FORCE_MAG)
        cos, sin = np.cos(theta), np.sin(theta)
        temp = (force + POLEMASS_LENGTH * theta_dot ** 2 * sin) / TOTAL_MASS
        theta_acc = (GRAVITY * sin - cos * temp) / (LENGTH * (4.0 / 3.0 - MASS_POLE * cos ** 2 / TOTAL_MASS))
        x_acc = temp - POLEMASS_LENGTH * theta_acc * cos / TOTAL_MASS
        self.state = np.stack([x + TAU * x_dot, x_dot + TAU * x_acc, theta + TAU * theta_dot, theta_dot + TAU * theta_acc], axis=1)
        self.ep_step += 1
        self.ep_ret += 1.0
        terminated = (np.abs(self.state[:, 0]) > X_LIMIT) | (np.abs(self.state[:, 2]) > THETA_LIMIT)
        truncated = (self.ep_step >= MAX_STEPS) & ~terminated
        done = terminated | truncated
        reward = np.ones(self.nr_envs, np.float32)
        final_obs = self.state.astype(np.float32)
        info = {"final_observation": final_obs, "done": done, "episode_return": self.ep_ret.copy(),
                "episode_length": self.ep_step.astype(np.float64)}
        if done.any():
            self._reset_rows(done)
            self.ep_step[done] = 0
            self.ep_ret[done] = 0.0
        return self.state.astype(np.float32), reward, terminated, truncated, info

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

    def get_single_action_logit_size(self):
        return self.single_action_space.n
