from rlx_amd.algorithms.algorithm_manager import extract_algorithm_name_from_file, register_algorithm
from rlx_amd.algorithms.ppo_gru.hip.ppo_gru import PPO_GRU
from rlx_amd.algorithms.ppo_gru.hip.default_config import get_config
from rlx_amd.algorithms.ppo_gru.hip.general_properties import GeneralProperties


PPO_GRU_HIP = extract_algorithm_name_from_file(__file__)
register_algorithm(PPO_GRU_HIP, get_config, PPO_GRU, GeneralProperties)
