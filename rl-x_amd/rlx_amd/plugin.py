"""Helpers the algorithm / environment plugins of this package are written with.

RL-X discovers plugins through three things per package directory (rl_x/algorithms/README.md:63-69,
rl_x/runner/runner.py:232-247): a default-config factory, a properties class the Runner's compatibility check reads
(runner.py:86-105), and a registration call at import time.  The helpers below build those from compact declarations, so a
plugin states WHAT it supports and which flags it has instead of repeating the boilerplate."""
from rlx_amd.algorithms.deep_learning_framework_type import DeepLearningFrameworkType
from rlx_amd.environments.action_space_type import ActionSpaceType
from rlx_amd.environments.data_interface_type import DataInterfaceType
from rlx_amd.environments.observation_space_type import ObservationSpaceType
from rlx_amd.environments.simulation_type import SimulationType
from rlx_amd.runner.config_dict import ConfigDict


def algorithm_properties(*, observations, actions, interfaces, framework):
    """Properties class of an algorithm plugin: which observation / action spaces and data interfaces it accepts (enum
    member names) and the framework branch a genuine rl_x Runner should take for it."""
    return type("GeneralProperties", (), {
        "observation_space_types": [ObservationSpaceType[n] for n in observations],
        "action_space_types": [ActionSpaceType[n] for n in actions],
        "data_interface_types": [DataInterfaceType[n] for n in interfaces],
        "deep_learning_framework_type": DeepLearningFrameworkType[framework],
    })


def environment_properties(*, observation, action, interface, simulation="DEFAULT"):
    """Properties class of an environment plugin (one value each)."""
    return type("GeneralProperties", (), {
        "observation_space_type": ObservationSpaceType[observation],
        "action_space_type": ActionSpaceType[action],
        "data_interface_type": DataInterfaceType[interface],
        "simulation_type": SimulationType[simulation],
    })


def flag_namespace(name, defaults):
    """ConfigDict with `name` first (omitted for name None: the runner namespace has none), then the flags in declaration
    order (the order `show_config` prints)."""
    config = ConfigDict()
    if name is not None:
        config.name = name
    for key, value in defaults.items():
        config[key] = value
    return config


def register_algorithm_plugin(init_file, get_config, model_class, properties):
    """Registers the algorithm under the name its directory spells (e.g. .../algorithms/ppo/hip -> "ppo.hip")."""
    from rlx_amd.algorithms.algorithm_manager import extract_algorithm_name_from_file, register_algorithm
    name = extract_algorithm_name_from_file(init_file)
    register_algorithm(name, get_config, model_class, properties)
    return name


def register_environment_plugin(init_file, get_config, create_train_and_eval_env, properties):
    from rlx_amd.environments.environment_manager import extract_environment_name_from_file, register_environment
    name = extract_environment_name_from_file(init_file)
    register_environment(name, get_config, create_train_and_eval_env, properties)
    return name
