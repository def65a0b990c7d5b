"""`synthetic.random_obs`: the device-resident benchmark environment (BASELINE.json configs[1..4] shapes)."""
from rlx_amd.plugin import register_environment_plugin
from . import create_env, default_config, general_properties

SYNTHETIC_RANDOM_OBS = register_environment_plugin(__file__, default_config.get_config, create_env.create_train_and_eval_env,
                                                   general_properties.GeneralProperties)
