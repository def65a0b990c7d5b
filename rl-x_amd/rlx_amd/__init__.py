"""rlx_amd -- MI355X-native PPO/SAC training hot path behind RL-X's plugin API.

Layout mirrors the reference package so the plugins drop into its Runner:
    rlx_amd.algorithms.*      registry + enums + `ppo.hip` (reference: rl_x/algorithms/)
    rlx_amd.environments.*    registry + enums + `synthetic.random_obs` (reference: rl_x/environments/)
    rlx_amd.runner            Runner / RunnerMode  (reference: rl_x/runner/)
    rlx_amd.hip               ctypes binding of librlxhip.so (include/rlx_hip.h)
"""
__version__ = "0.1.0"
