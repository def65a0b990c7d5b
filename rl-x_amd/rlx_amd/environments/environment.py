"""Registry record of an environment plugin (field names as rl_x/environments/environment.py:1-6)."""
from collections import namedtuple

Environment = namedtuple("Environment", ["name", "get_default_config", "create_train_and_eval_env", "general_properties"])
