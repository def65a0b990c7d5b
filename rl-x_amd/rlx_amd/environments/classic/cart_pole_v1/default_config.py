"""Flag namespace as rl_x/environments/gym/classic/cart_pole_v1/default_config.py (nr_envs defaults to the 8 of
BASELINE.json configs[0])."""
from rlx_amd.runner.config_dict import ConfigDict


def get_config(environment_name):
    config = ConfigDict()

    config.name = environment_name

    config.type = "CartPole-v1"
    config.seed = 1
    config.nr_envs = 8
    config.render = False
    config.copy_train_env_for_eval = True
    config.device = "cpu"            # the simulation runs on the host

    return config
