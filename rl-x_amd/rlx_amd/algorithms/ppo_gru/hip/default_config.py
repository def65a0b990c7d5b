"""`ppo_gru.hip` flags: keys and defaults of the reference's fully-jitted GRU PPO
(rl_x/algorithms/ppo_gru/flax_full_jit/default_config.py:9-33) plus `threefry_partitionable` (JAX's jax_threefry_partitionable)."""
from rlx_amd.plugin import flag_namespace

FLAGS = dict(
    device="gpu", nr_parallel_seeds=1, total_timesteps=2e9,
    learning_rate=4e-4, anneal_learning_rate=True, nr_steps=128, nr_epochs=10, minibatch_size=32768,
    gamma=0.99, gae_lambda=0.9, clip_range=0.1, entropy_coef=0.0, critic_coef=1.0, max_grad_norm=5.0, std_dev=1.0,
    # recurrent policy
    obs_encoding_dim=128, gru_hidden_dim=64,
    gru_obs_combine_method="concat",        # or "film" (policy.py:95-100)
    share_gru_obs_encoder=False,
    action_clipping_and_rescaling=False, evaluation_and_save_frequency=17301504, evaluation_active=False,
    threefry_partitionable=True,
)


def get_config(algorithm_name):
    return flag_namespace(algorithm_name, FLAGS)
