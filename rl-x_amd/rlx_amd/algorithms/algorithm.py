"""Registry record of an algorithm plugin (field names as rl_x/algorithms/algorithm.py:1-6: the Runner reads them)."""
from collections import namedtuple

Algorithm = namedtuple("Algorithm", ["name", "get_default_config", "get_model_class", "general_properties"])
