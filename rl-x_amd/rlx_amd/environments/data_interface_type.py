"""rl_x/environments/data_interface_type.py (enum identity shared with a genuine rl_x when present)."""
try:
    from rl_x.environments.data_interface_type import DataInterfaceType  # noqa: F401
except ImportError:
    from enum import Enum

    class DataInterfaceType(Enum):
        LIST = 0
        NUMPY = 1
        TORCH = 2
        JAX = 3
