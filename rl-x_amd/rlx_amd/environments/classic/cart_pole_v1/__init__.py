"""`classic.cart_pole_v1`: CartPole-v1 as a host numpy vector env (BASELINE.json configs[0])."""
from rlx_amd.plugin import register_environment_plugin
from . import create_env, default_config, general_properties

CLASSIC_CART_POLE_V1 = register_environment_plugin(__file__, default_config.get_config, create_env.create_train_and_eval_env,
                                                   general_properties.GeneralProperties)
