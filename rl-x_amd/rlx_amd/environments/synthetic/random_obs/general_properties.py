"""Device-resident synthetic env.  simulation DEFAULT (not WARP / ISAAC_LAB / MANISKILL): a genuine rl_x Runner then skips its
algorithm / environment device-equality probe (rl_x/runner/runner.py:102,116-128)."""
from rlx_amd.plugin import environment_properties

GeneralProperties = environment_properties(observation="FLAT_VALUES", action="CONTINUOUS", interface="TORCH")
