"""What `sac.hip` can be paired with (a discrete environment makes the Runner raise "Incompatible action space type")."""
from rlx_amd.plugin import algorithm_properties

GeneralProperties = algorithm_properties(observations=("FLAT_VALUES",), actions=("CONTINUOUS",), interfaces=("TORCH",),
                                         framework="TORCH")
