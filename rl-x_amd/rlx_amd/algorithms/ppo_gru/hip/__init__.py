"""`ppo_gru.hip`: recurrent PPO with a GRU cell (same loop and kernels as ppo_lstm.hip)."""
from rlx_amd.plugin import register_algorithm_plugin
from . import default_config, general_properties
from .ppo_gru import PPO_GRU

PPO_GRU_HIP = register_algorithm_plugin(__file__, default_config.get_config, PPO_GRU, general_properties.GeneralProperties)
