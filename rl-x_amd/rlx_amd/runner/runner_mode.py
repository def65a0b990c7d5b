"""rl_x/runner/runner_mode.py:1-4."""


class RunnerMode:
    TRAIN = "train"
    TEST = "test"
    SHOW_CONFIG = "show_config"
