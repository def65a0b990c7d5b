"""Runner flag namespace: keys and defaults of rl_x/runner/default_config.py:5-29.  The four jax_* keys are accepted for
command-line compatibility and ignored (no JAX here)."""
import time

from rlx_amd.plugin import flag_namespace


def get_config(runner_mode):
    config = flag_namespace(None, dict(
        mode=runner_mode,
        # sinks
        track_console=False, track_tb=False, track_wandb=False,
        wandb_entity="placeholder", project_name="placeholder", exp_name="placeholder",
        run_name=f"{int(time.time())}", notes="placeholder",
        # checkpoints
        save_model=False, load_model="",
        nr_test_episodes=10,          # --runner.mode=test
        jax_cache_dir="/tmp/jax_cache", jax_default_matmul_precision="bfloat16",
        jax_exec_time_optimization_effort=0.0, jax_memory_fitting_effort=1.0))
    return config
