"""Same shape as rl_x/environments/custom_mujoco/ant/warp_torch/create_env.py:6-17."""
from rlx_amd.environments.synthetic.random_obs.environment import RandomObsEnv
from rlx_amd.environments.synthetic.random_obs.general_properties import GeneralProperties


def create_train_and_eval_env(config):
    train_env = RandomObsEnv(config.environment)
    train_env.general_properties = GeneralProperties

    if config.environment.copy_train_env_for_eval:
        return train_env, train_env

    eval_env = RandomObsEnv(config.environment, eval_stream=True)
    eval_env.general_properties = GeneralProperties

    return train_env, eval_env
