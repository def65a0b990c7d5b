"""Flag namespace of the synthetic env; the generic keys follow the reference's vector-sim envs
(rl_x/environments/custom_mujoco/ant/mjx/default_config.py:7-15)."""
from rlx_amd.runner.config_dict import ConfigDict


def get_config(environment_name):
    config = ConfigDict()

    config.name = environment_name

    config.seed = 1
    config.nr_envs = 4096            # GLOBAL number of envs; sharded over ranks when torch.distributed is up
    config.render = False
    config.device = "gpu"
    config.horizon = 1000
    config.copy_train_env_for_eval = True

    config.obs_dim = 17              # HalfCheetah-shaped (BASELINE.json configs[1])
    config.act_dim = 6
    config.termination_probability = 1e-3
    config.reward_noise = 0.1

    return config
