"""The three `--runner.mode` values (rl_x/runner/runner_mode.py)."""


class RunnerMode:
    TRAIN, TEST, SHOW_CONFIG = "train", "test", "show_config"
