from rlx_amd.environments.action_space_type import ActionSpaceType
from rlx_amd.environments.observation_space_type import ObservationSpaceType
from rlx_amd.environments.data_interface_type import DataInterfaceType
from rlx_amd.environments.simulation_type import SimulationType


class GeneralProperties:   # the reference's rl_x/environments/gym/classic/cart_pole_v1/general_properties.py
    observation_space_type = ObservationSpaceType.FLAT_VALUES
    action_space_type = ActionSpaceType.DISCRETE
    data_interface_type = DataInterfaceType.NUMPY
    simulation_type = SimulationType.DEFAULT
