"""`fastsac.hip` flags = rl_x/algorithms/fastsac/pytorch/default_config.py:9-38.  `compile_mode` has no meaning here (nothing is
traced); `bf16_mixed_precision_training` defaults to False: the library computes in fp32 (True is refused, not emulated)."""
from rlx_amd.plugin import flag_namespace

FLAGS = dict(
    device="gpu", compile_mode="none", bf16_mixed_precision_training=False, total_timesteps=2000158720, learning_rate=3e-4,
    anneal_learning_rate=False, weight_decay=0.001, adam_beta1=0.9, adam_beta2=0.95,
    # replay
    batch_size=8192, buffer_size_per_env=1024, learning_starts=10, n_steps=1,
    # objective
    v_min=-20.0, v_max=20.0, tau=0.125, gamma=0.97, nr_atoms=101, target_entropy=0.0, alpha_init=0.001, log_std_min=-5.0, log_std_max=0.0,
    nr_critic_updates_per_policy_update=4, nr_policy_updates_per_step=2, clipped_double_q_learning=False, max_grad_norm=-1.0,
    enable_observation_normalization=True,
    logging_frequency=40960, evaluation_frequency=-1, save_frequency=4096000,
    threefry_partitionable=True,
)


def get_config(algorithm_name):
    return flag_namespace(algorithm_name, FLAGS)
