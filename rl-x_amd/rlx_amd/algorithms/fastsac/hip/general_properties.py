"""What `fastsac.hip` can be paired with (rl_x/algorithms/fastsac/pytorch/general_properties.py)."""
from rlx_amd.plugin import algorithm_properties

GeneralProperties = algorithm_properties(observations=("FLAT_VALUES",), actions=("CONTINUOUS",), interfaces=("TORCH",),
                                         framework="TORCH")
