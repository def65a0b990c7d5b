"""`ppo.hip` flag namespace.  Keys/defaults are those of the reference's fully-jitted PPO
(rl_x/algorithms/ppo/flax_full_jit/default_config.py:9-26) -- the configuration BASELINE.json
is quoted on -- plus `network_architecture` / `nr_hidden_units` to select the host-loop variant's
256-256 tanh nets (rl_x/algorithms/ppo/flax/default_config.py:24) and `threefry_partitionable`
(JAX's `jax_threefry_partitionable`, default True since JAX 0.5.0)."""
from rlx_amd.runner.config_dict import ConfigDict


def get_config(algorithm_name):
    config = ConfigDict()

    config.name = algorithm_name

    config.device = "gpu"
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
    config.action_clipping_and_rescaling = False
    config.evaluation_frequency = -1
    config.evaluation_episodes = 10

    config.network_architecture = "full_jit"   # "full_jit": 512(LN)-256-128 ELU; "flax": H-H tanh
    config.nr_hidden_units = 256                # used by network_architecture="flax"
    config.threefry_partitionable = True
    config.force_distributed_update = False     # run the multi-GPU update protocol even with one rank (test aid)
    config.fused_rollout = True                 # one kernel per acting step when the network shapes allow it

    return config
