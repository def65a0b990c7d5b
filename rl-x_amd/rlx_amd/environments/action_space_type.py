"""rl_x/environments/action_space_type.py (enum identity shared with a genuine rl_x when present)."""
try:
    from rl_x.environments.action_space_type import ActionSpaceType  # noqa: F401
except ImportError:
    from enum import Enum

    class ActionSpaceType(Enum):
        CONTINUOUS = 0
        DISCRETE = 1
