"""Flag namespace of the host-side synthetic env (generic keys as rl_x/environments/gym/mujoco/humanoid_v4/default_config.py)."""
from rlx_amd.runner.config_dict import ConfigDict


def get_config(environment_name):
    config = ConfigDict()

    config.name = environment_name

    config.seed = 1
    config.nr_envs = 64
    config.render = False
    config.device = "cpu"            # the simulation runs on the host: observations cross PCIe every step
    config.horizon = 1000

    config.obs_dim = 17
    config.act_dim = 6
    config.termination_probability = 1e-3
    config.reward_noise = 0.1

    return config
