from rlx_amd.algorithms.algorithm_manager import extract_algorithm_name_from_file, register_algorithm
from rlx_amd.algorithms.ppo.hip.ppo import PPO
from rlx_amd.algorithms.ppo.hip.default_config import get_config
from rlx_amd.algorithms.ppo.hip.general_properties import GeneralProperties


PPO_HIP = extract_algorithm_name_from_file(__file__)
register_algorithm(PPO_HIP, get_config, PPO, GeneralProperties)
