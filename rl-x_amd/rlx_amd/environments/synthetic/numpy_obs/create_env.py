"""Same shape as rl_x/environments/gym/mujoco/humanoid_v4/create_env.py (train env + separate eval env)."""
from rlx_amd.environments.synthetic.numpy_obs.environment import NumpyObsEnv
from rlx_amd.environments.synthetic.numpy_obs.general_properties import GeneralProperties


def create_train_and_eval_env(config):
    train_env = NumpyObsEnv(config.environment)
    train_env.general_properties = GeneralProperties
    eval_env = NumpyObsEnv(config.environment, eval_stream=True)
    eval_env.general_properties = GeneralProperties
    return train_env, eval_env
