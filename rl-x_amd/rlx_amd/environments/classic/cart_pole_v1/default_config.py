"""Flags as rl_x/environments/gym/classic/cart_pole_v1/default_config.py; nr_envs defaults to the 8 of BASELINE.json configs[0]."""
from rlx_amd.plugin import flag_namespace

FLAGS = dict(type="CartPole-v1", seed=1, nr_envs=8, render=False, copy_train_env_for_eval=True,
             device="cpu")           # the simulation runs on the host


def get_config(environment_name):
    return flag_namespace(environment_name, FLAGS)
