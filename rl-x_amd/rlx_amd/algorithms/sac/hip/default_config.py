"""`sac.hip` flags = rl_x/algorithms/sac/flax/default_config.py:9-24 (+ threefry_partitionable)."""
from rlx_amd.plugin import flag_namespace

FLAGS = dict(
    device="gpu", total_timesteps=1e9, learning_rate=3e-4, anneal_learning_rate=False,
    # replay
    buffer_size=1e6, learning_starts=5000, batch_size=256,
    # objective
    tau=0.005, gamma=0.99, target_entropy="auto", log_std_min=-20.0, log_std_max=2.0,
    nr_hidden_units=256,
    logging_frequency=3000, evaluation_frequency=-1, evaluation_episodes=10,
    threefry_partitionable=True,
)


def get_config(algorithm_name):
    return flag_namespace(algorithm_name, FLAGS)
