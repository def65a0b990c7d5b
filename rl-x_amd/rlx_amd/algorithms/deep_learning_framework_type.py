"""rl_x/algorithms/deep_learning_framework_type.py:4-6 (enum identity is shared with a
genuine rl_x when present: the runner compares members by identity, runner.py:86-91)."""
try:
    from rl_x.algorithms.deep_learning_framework_type import DeepLearningFrameworkType  # noqa: F401
except ImportError:
    from enum import Enum

    class DeepLearningFrameworkType(Enum):
        TORCH = 0
        JAX = 1
