"""Flags of the device-resident synthetic env; the generic ones follow the reference's vector-sim envs
(rl_x/environments/custom_mujoco/ant/mjx/default_config.py:7-15)."""
from rlx_amd.plugin import flag_namespace

FLAGS = dict(
    seed=1,
    nr_envs=4096,                    # GLOBAL number of envs; sharded over ranks when torch.distributed is up
    render=False, device="gpu", horizon=1000, copy_train_env_for_eval=True,
    obs_dim=17, act_dim=6,           # HalfCheetah-shaped (BASELINE.json configs[1])
    termination_probability=1e-3, reward_noise=0.1,
)


def get_config(environment_name):
    return flag_namespace(environment_name, FLAGS)
