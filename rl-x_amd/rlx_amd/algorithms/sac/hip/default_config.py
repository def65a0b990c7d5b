"""`sac.hip` flags = rl_x/algorithms/sac/flax/default_config.py:9-24 (+ threefry_partitionable, + FastSAC's
enable_observation_normalization)."""
from rlx_amd.plugin import flag_namespace

FLAGS = dict(
    device="gpu", total_timesteps=1e9, learning_rate=3e-4, anneal_learning_rate=False,
    # replay
    buffer_size=1e6, learning_starts=5000, batch_size=256,
    # objective
    tau=0.005, gamma=0.99, target_entropy="auto", log_std_min=-20.0, log_std_max=2.0,
    nr_hidden_units=256,
    # "flax": sac/flax (2 x Dense(nr_hidden_units) + ReLU, noise keys split(key, 2B+1), replay indices from numpy's Generator
    # on the host); "full_jit": sac/flax_full_jit (512-LayerNorm-256-128 ELU nets, keys split(key, 2B+2), replay indices
    # drawn on the device from keys[1])
    network_architecture="flax",
    # FastSAC's running observation normaliser (fastsac/pytorch/default_config.py: enable_observation_normalization)
    enable_observation_normalization=False,
    logging_frequency=3000, evaluation_frequency=-1, evaluation_episodes=10,
    threefry_partitionable=True,
)


def get_config(algorithm_name):
    return flag_namespace(algorithm_name, FLAGS)
