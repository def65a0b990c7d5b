"""`sac.hip`: SAC whose update runs in librlxhip.so."""
from rlx_amd.plugin import register_algorithm_plugin
from . import default_config, general_properties
from .sac import SAC

SAC_HIP = register_algorithm_plugin(__file__, default_config.get_config, SAC, general_properties.GeneralProperties)
