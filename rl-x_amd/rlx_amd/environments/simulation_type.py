"""rl_x/environments/simulation_type.py (enum identity shared with a genuine rl_x when present)."""
try:
    from rl_x.environments.simulation_type import SimulationType  # noqa: F401
except ImportError:
    from enum import Enum

    class SimulationType(Enum):
        DEFAULT = 0
        JAX_BASED = 1
        ISAAC_LAB = 2
        MANISKILL = 3
        WARP = 4
