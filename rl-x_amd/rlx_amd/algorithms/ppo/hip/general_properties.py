"""What `ppo.hip` can be paired with.  NUMPY: host environments through pinned staging buffers.  DISCRETE: the Categorical
head (2..8 actions).  framework TORCH: a genuine rl_x Runner then takes its torch branch and skips all JAX set-up
(rl_x/runner/runner.py:108-174) -- PyTorch-ROCm only stores the tensors here."""
from rlx_amd.plugin import algorithm_properties

GeneralProperties = algorithm_properties(observations=("FLAT_VALUES",), actions=("CONTINUOUS", "DISCRETE"),
                                         interfaces=("TORCH", "NUMPY"), framework="TORCH")
