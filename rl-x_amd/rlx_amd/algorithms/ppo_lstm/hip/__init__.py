"""`ppo_lstm.hip`: recurrent PPO (LSTM cell) on the HIP kernels."""
from rlx_amd.plugin import register_algorithm_plugin
from . import default_config, general_properties
from .ppo_lstm import PPO_LSTM

PPO_LSTM_HIP = register_algorithm_plugin(__file__, default_config.get_config, PPO_LSTM, general_properties.GeneralProperties)
