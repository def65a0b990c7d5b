from rlx_amd.environments.action_space_type import ActionSpaceType
from rlx_amd.environments.observation_space_type import ObservationSpaceType
from rlx_amd.environments.data_interface_type import DataInterfaceType
from rlx_amd.algorithms.deep_learning_framework_type import DeepLearningFrameworkType


class GeneralProperties:
    observation_space_types = [ObservationSpaceType.FLAT_VALUES]
    action_space_types = [ActionSpaceType.CONTINUOUS, ActionSpaceType.DISCRETE]   # DISCRETE: Categorical head, 2..8 actions
    data_interface_types = [DataInterfaceType.TORCH, DataInterfaceType.NUMPY]   # NUMPY: host envs through pinned staging

    # TORCH: a genuine rl_x Runner then takes its torch branch and skips all JAX setup
    # (rl_x/runner/runner.py:108-174); PyTorch-ROCm only stores the tensors here.
    deep_learning_framework_type = DeepLearningFrameworkType.TORCH
