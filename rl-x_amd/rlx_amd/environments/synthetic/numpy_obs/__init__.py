"""`synthetic.numpy_obs`: the same task simulated on the host (NUMPY data interface)."""
from rlx_amd.plugin import register_environment_plugin
from . import create_env, default_config, general_properties

SYNTHETIC_NUMPY_OBS = register_environment_plugin(__file__, default_config.get_config, create_env.create_train_and_eval_env,
                                                  general_properties.GeneralProperties)
