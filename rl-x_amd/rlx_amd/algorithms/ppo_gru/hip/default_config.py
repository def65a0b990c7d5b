"""`ppo_gru.hip` flag namespace: keys/defaults of the reference's fully-jitted GRU PPO
(rl_x/algorithms/ppo_gru/flax_full_jit/default_config.py:9-33) plus `threefry_partitionable`
(JAX's `jax_threefry_partitionable`)."""
from rlx_amd.runner.config_dict import ConfigDict


def get_config(algorithm_name):
    config = ConfigDict()

    config.name = algorithm_name

    config.device = "gpu"
    config.nr_parallel_seeds = 1
    config.total_timesteps = 2e9
    config.learning_rate = 4e-4
    config.anneal_learning_rate = True
    config.nr_steps = 128
    config.nr_epochs = 10
    config.minibatch_size = 32768
    config.gamma = 0.99
    config.gae_lambda = 0.9
    config.clip_range = 0.1
    config.entropy_coef = 0.0
    config.critic_coef = 1.0
    config.max_grad_norm = 5.0
    config.std_dev = 1.0
    config.obs_encoding_dim = 128
    config.gru_hidden_dim = 64
    config.gru_obs_combine_method = "concat"   # "film" is not built: raises
    config.share_gru_obs_encoder = False
    config.action_clipping_and_rescaling = False
    config.evaluation_and_save_frequency = 17301504
    config.evaluation_active = False

    config.threefry_partitionable = True

    return config
