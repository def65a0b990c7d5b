"""Host-side synthetic env: observations cross PCIe every step (NUMPY data interface)."""
from rlx_amd.plugin import environment_properties

GeneralProperties = environment_properties(observation="FLAT_VALUES", action="CONTINUOUS", interface="NUMPY")
