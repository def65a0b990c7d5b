"""ppo_gru.hip -- PPO with a GRU policy (rl_x/algorithms/ppo_gru/flax_full_jit): the same training loop, acting step,
sequence minibatches and torso as ppo_lstm.hip with flax.linen.GRUCell as the recurrent cell and a single carry h
(policy.py:52,68-69,112-121).  Kernels: k_gru_seq_fwd / k_gru_seq_bwd (rl-x_amd/csrc/lstm_kernels.h) behind the
rlx_ppo_lstm_* entry points with `rlx_lstm_policy_desc.cell = RLX_CELL_GRU`."""
from rlx_amd.algorithms.ppo_lstm.hip.ppo_lstm import PPO_LSTM
from rlx_amd.algorithms.ppo_gru.hip.general_properties import GeneralProperties


class PPO_GRU(PPO_LSTM):
    CELL = "gru"

    def general_properties():
        return GeneralProperties
