"""`fastsac.hip`: FastSAC (distributional twin critics, LayerNorm + SiLU networks, AdamW) whose update runs in librlxhip.so."""
from rlx_amd.plugin import register_algorithm_plugin
from . import default_config, general_properties
from .fastsac import FastSAC

FASTSAC_HIP = register_algorithm_plugin(__file__, default_config.get_config, FastSAC, general_properties.GeneralProperties)
