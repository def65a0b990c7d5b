"""Environment registry -- same functions as rl_x/environments/environment_manager.py:5-25;
shares the genuine registry when `rl_x` is importable (see algorithm_manager.py)."""
from os import sep as slash

try:
    from rl_x.environments.environment_manager import (  # noqa: F401
        register_environment, get_environment_config, get_environment_create_train_and_eval_env,
        get_environment_general_properties)
    USING_REFERENCE_REGISTRY = True
except ImportError:
    from rlx_amd.environments.environment import Environment

    USING_REFERENCE_REGISTRY = False
    _environments = {}

    def register_environment(name, get_default_config, create_train_and_eval_env, general_properties):
        _environments[name] = Environment(name, get_default_config, create_train_and_eval_env, general_properties)

    def get_environment_config(environment_name):
        return _environments[environment_name].get_default_config(environment_name)

    def get_environment_create_train_and_eval_env(environment_name):
        return _environments[environment_name].create_train_and_eval_env

    def get_environment_general_properties(environment_name):
        return _environments[environment_name].general_properties


def extract_environment_name_from_file(file_name):
    return file_name.split(f"environments{slash}")[1].split(f"{slash}__init__.py")[0].replace(slash, ".")
