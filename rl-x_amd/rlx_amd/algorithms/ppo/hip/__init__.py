"""`ppo.hip`: PPO whose iteration runs in librlxhip.so (Gaussian or Categorical policy)."""
from rlx_amd.plugin import register_algorithm_plugin
from . import default_config, general_properties
from .ppo import PPO

PPO_HIP = register_algorithm_plugin(__file__, default_config.get_config, PPO, general_properties.GeneralProperties)
