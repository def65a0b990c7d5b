"""Algorithm registry -- same functions and semantics as
rl_x/algorithms/algorithm_manager.py:5-25 (module-level dict, last registration wins).

Shim: when a genuine `rl_x` package is importable the functions below ARE its registry
functions, so plugins of this package register into the reference Runner unchanged
(`Runner(implementation_package_names=["rl_x", "rlx_amd"])`, runner.py:232-247).
"""
from os import sep as slash

try:  # genuine RL-X present: share its registry
    from rl_x.algorithms.algorithm_manager import (  # noqa: F401
        register_algorithm, get_algorithm_config, get_algorithm_model_class, get_algorithm_general_properties)
    USING_REFERENCE_REGISTRY = True
except ImportError:
    from rlx_amd.algorithms.algorithm import Algorithm

    USING_REFERENCE_REGISTRY = False
    _algorithms = {}

    def register_algorithm(name, get_default_config, get_model_class, general_properties):
        _algorithms[name] = Algorithm(name, get_default_config, get_model_class, general_properties)

    def get_algorithm_config(algorithm_name):
        return _algorithms[algorithm_name].get_default_config(algorithm_name)

    def get_algorithm_model_class(algorithm_name):
        return _algorithms[algorithm_name].get_model_class

    def get_algorithm_general_properties(algorithm_name):
        return _algorithms[algorithm_name].general_properties


def extract_algorithm_name_from_file(file_name):
    """algorithm_manager.py:8-9: path after `algorithms/` with separators -> dots."""
    return file_name.split(f"algorithms{slash}")[1].split(f"{slash}__init__.py")[0].replace(slash, ".")
