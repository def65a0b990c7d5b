"""CartPole-v1 on the host: four observations, two discrete actions."""
from rlx_amd.plugin import environment_properties

GeneralProperties = environment_properties(observation="FLAT_VALUES", action="DISCRETE", interface="NUMPY")
