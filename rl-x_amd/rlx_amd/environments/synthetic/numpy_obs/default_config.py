"""Flags of the host-side synthetic env (generic keys as rl_x/environments/gym/mujoco/humanoid_v4/default_config.py)."""
from rlx_amd.plugin import flag_namespace

FLAGS = dict(
    seed=1, nr_envs=64, render=False,
    device="cpu",                    # the simulation runs on the host: observations cross PCIe every step
    horizon=1000, obs_dim=17, act_dim=6, termination_probability=1e-3, reward_noise=0.1,
)


def get_config(environment_name):
    return flag_namespace(environment_name, FLAGS)
