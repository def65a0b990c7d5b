"""rl_x/environments/observation_space_type.py (enum identity shared with a genuine rl_x when present)."""
try:
    from rl_x.environments.observation_space_type import ObservationSpaceType  # noqa: F401
except ImportError:
    from enum import Enum

    class ObservationSpaceType(Enum):
        FLAT_VALUES = 0
        IMAGES = 1
