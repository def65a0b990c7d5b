"""Same shape as rl_x/environments/gym/classic/cart_pole_v1/create_env.py (train env + eval env)."""
from rlx_amd.environments.classic.cart_pole_v1.environment import CartPoleVecEnv
from rlx_amd.environments.classic.cart_pole_v1.general_properties import GeneralProperties


def create_train_and_eval_env(config):
    train_env = CartPoleVecEnv(config.environment)
    train_env.general_properties = GeneralProperties
    eval_env = CartPoleVecEnv(config.environment, eval_stream=True)
    eval_env.general_properties = GeneralProperties
    return train_env, eval_env
