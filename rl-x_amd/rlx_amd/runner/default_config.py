"""Runner flag namespace; keys and defaults of rl_x/runner/default_config.py:5-29 (the
four jax_* keys are accepted for command-line compatibility and ignored: no JAX here)."""
import time

from rlx_amd.runner.config_dict import ConfigDict


def get_config(runner_mode):
    config = ConfigDict()

    config.mode = runner_mode

    config.track_console = False
    config.track_tb = False
    config.track_wandb = False
    config.wandb_entity = "placeholder"
    config.project_name = "placeholder"
    config.exp_name = "placeholder"
    config.run_name = f"{int(time.time())}"
    config.notes = "placeholder"

    config.save_model = False
    config.load_model = ""

    config.nr_test_episodes = 10  # if runner mode = test

    config.jax_cache_dir = "/tmp/jax_cache"
    config.jax_default_matmul_precision = "bfloat16"
    config.jax_exec_time_optimization_effort = 0.0
    config.jax_memory_fitting_effort = 1.0

    return config
