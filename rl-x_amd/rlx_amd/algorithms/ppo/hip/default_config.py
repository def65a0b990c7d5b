"""`ppo.hip` flags.  Keys and defaults follow the reference's fully-jitted PPO
(rl_x/algorithms/ppo/flax_full_jit/default_config.py:9-26, the configuration BASELINE.json is quoted on) so existing command
lines keep working; the last group is this build's own."""
from rlx_amd.plugin import flag_namespace

PPO_FLAGS = dict(
    device="gpu", total_timesteps=2e9,
    # optimisation
    learning_rate=4e-4, anneal_learning_rate=True, nr_steps=128, nr_epochs=10, minibatch_size=32768,
    # objective
    gamma=0.99, gae_lambda=0.9, clip_range=0.1, entropy_coef=0.0, critic_coef=1.0, max_grad_norm=5.0,
    # policy
    std_dev=1.0, action_clipping_and_rescaling=False,
    evaluation_frequency=-1, evaluation_episodes=10,
    # ---- this build
    network_architecture="full_jit",   # "full_jit": 512(LN)-256-128 ELU; "flax": H-H tanh (rl_x/algorithms/ppo/flax/default_config.py:24)
    nr_hidden_units=256,               # H of network_architecture="flax"
    threefry_partitionable=True,       # JAX's jax_threefry_partitionable (default True since JAX 0.5.0)
    force_distributed_update=False,    # run the multi-GPU update protocol even with one rank (test aid)
    fused_rollout=True,                # one kernel per acting step when the network shapes allow it
)


def get_config(algorithm_name):
    return flag_namespace(algorithm_name, PPO_FLAGS)
