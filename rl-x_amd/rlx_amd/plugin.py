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



class MetricSink:
    """Where a plugin's scalars of one logging step go: the console table, a TensorBoard writer, a wandb record -- the three
    sinks the Runner can switch on (`--runner.track_console / track_tb / track_wandb`, rl_x/runner/default_config.py:9-11).
    One sink object per model; `rank` != 0 swallows everything (data-parallel jobs log once).

        sink.open(step); sink.put(name, value) ...; sink.close()
    """
    NAME_WIDTH, VALUE_WIDTH = 30, 14

    def __init__(self, logger, writer, *, console, tensorboard, wandb, rank=0):
        self.logger, self.writer = logger, writer
        self.console, self.tensorboard, self.wandb = bool(console), bool(tensorboard), bool(wandb)
        self.active = rank == 0
        self.step, self.record = 0, None

    def _rule(self, left, mid, right):
        return left + "\u2500" * (self.NAME_WIDTH + 1) + mid + "\u2500" * (self.VALUE_WIDTH + 2) + right

    def open(self, step):
        if not self.active:
            return
        self.step = int(step)
        self.record = {"global_step": self.step} if self.wandb else None
        self.logger.info(self._rule("\u250c", "\u252c", "\u2510") if self.console else f"Step: {step}")

    def put(self, name, value):
        if not self.active:
            return
        if self.record is not None:
            self.record[name] = value
        if self.tensorboard:
            self.writer.add_scalar(name, value, self.step)
        if self.console:
            import numpy as np
            text = str(np.format_float_positional(value, trim="-"))[:self.VALUE_WIDTH]
            self.logger.info(f"\u2502 {name:<{self.NAME_WIDTH}}\u2502 {text:<{self.VALUE_WIDTH}} \u2502")

    def close(self, commit=True):
        if not self.active:
            return
        if self.record is not None:
            import wandb
            wandb.log(self.record, commit=commit)
        if self.console:
            self.logger.info(self._rule("\u2514", "\u2534", "\u2518"))

    def write(self, step, scalars):
        """One whole logging step: every (name, value) of `scalars` in order."""
        self.open(step)
        for name, value in scalars.items():
            self.put(name, value)
        self.close()


def adopt_checkpoint_config(config, stored_algorithm_config, explicitly_set):
    """Algorithm flags stored with a checkpoint win over the defaults, flags given on the command line win over both
    (the rule of the reference's `load`, rl_x/algorithms/ppo/flax/ppo.py:444-452)."""
    for key in list(config.algorithm.keys()):
        if key in stored_algorithm_config and f"algorithm.{key}" not in explicitly_set:
            config.algorithm[key] = stored_algorithm_config[key]
