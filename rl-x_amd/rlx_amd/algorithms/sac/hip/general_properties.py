from rlx_amd.environments.action_space_type import ActionSpaceType
from rlx_amd.environments.observation_space_type import ObservationSpaceType
from rlx_amd.environments.data_interface_type import DataInterfaceType
from rlx_amd.algorithms.deep_learning_framework_type import DeepLearningFrameworkType


class GeneralProperties:
    observation_space_types = [ObservationSpaceType.FLAT_VALUES]
    action_space_types = [ActionSpaceType.CONTINUOUS]
    data_interface_types = [DataInterfaceType.TORCH]

    deep_learning_framework_type = DeepLearningFrameworkType.TORCH
