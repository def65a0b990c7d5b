"""`sac.hip` flag namespace = rl_x/algorithms/sac/flax/default_config.py:9-24 (+ threefry_partitionable)."""
from rlx_amd.runner.config_dict import ConfigDict


def get_config(algorithm_name):
    config = ConfigDict()

    config.name = algorithm_name

    config.device = "gpu"
    config.total_timesteps = 1e9
    config.learning_rate = 3e-4
    config.anneal_learning_rate = False
    config.buffer_size = 1e6
    config.learning_starts = 5000
    config.batch_size = 256
    config.tau = 0.005
    config.gamma = 0.99
    config.target_entropy = "auto"
    config.log_std_min = -20.0
    config.log_std_max = 2.0
    config.nr_hidden_units = 256
    config.logging_frequency = 3000
    config.evaluation_frequency = -1
    config.evaluation_episodes = 10

    config.threefry_partitionable = True

    return config
