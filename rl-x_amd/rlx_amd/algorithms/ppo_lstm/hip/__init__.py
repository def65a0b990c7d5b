from rlx_amd.algorithms.algorithm_manager import extract_algorithm_name_from_file, register_algorithm
from rlx_amd.algorithms.ppo_lstm.hip.ppo_lstm import PPO_LSTM
from rlx_amd.algorithms.ppo_lstm.hip.default_config import get_config
from rlx_amd.algorithms.ppo_lstm.hip.general_properties import GeneralProperties


PPO_LSTM_HIP = extract_algorithm_name_from_file(__file__)
register_algorithm(PPO_LSTM_HIP, get_config, PPO_LSTM, GeneralProperties)
