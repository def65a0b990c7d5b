"""CPU: host helpers of the plugins (rlx_amd/plugin.py) that mirror pieces of the reference's algorithm classes:
the three metric sinks the Runner switches on (rl_x/runner/default_config.py:9-11; console table / TensorBoard / wandb as in
rl_x/algorithms/ppo/flax/ppo.py:372-397) and the precedence of algorithm flags when a checkpoint is loaded (ppo.py:444-452)."""
import sys
import types

from rlx_amd.plugin import MetricSink, adopt_checkpoint_config
from rlx_amd.runner.config_dict import ConfigDict


class _Log:
    def __init__(self):
        self.lines = []

    def info(self, s):
        self.lines.append(s)


class _Writer:
    def __init__(self):
        self.rows = []

    def add_scalar(self, name, value, step):
        self.rows.append((name, value, step))


def test_console_table_and_tensorboard_rows():
    log, w = _Log(), _Writer()
    sink = MetricSink(log, w, console=True, tensorboard=True, wandb=False)
    sink.write(4096, {"loss/critic_loss": 0.125, "time/sps": 5000000})
    assert w.rows == [("loss/critic_loss", 0.125, 4096), ("time/sps", 5000000, 4096)]
    assert len(log.lines) == 4 and log.lines[0].startswith("┌") and log.lines[-1].startswith("└")
    assert "loss/critic_loss" in log.lines[1] and "0.125" in log.lines[1] and "5000000" in log.lines[2]
    assert len({len(x) for x in log.lines}) == 1                      # a table: every line equally wide


def test_wandb_record_is_committed_once_per_step(monkeypatch):
    calls = []
    monkeypatch.setitem(sys.modules, "wandb", types.SimpleNamespace(log=lambda rec, commit=True: calls.append((dict(rec), commit))))
    sink = MetricSink(_Log(), None, console=False, tensorboard=False, wandb=True)
    sink.write(10, {"a": 1.0, "b": 2.0})
    assert calls == [({"global_step": 10, "a": 1.0, "b": 2.0}, True)]


def test_other_ranks_log_nothing():
    log, w = _Log(), _Writer()
    sink = MetricSink(log, w, console=True, tensorboard=True, wandb=False, rank=3)
    sink.write(1, {"a": 1.0})
    assert not log.lines and not w.rows


def test_checkpoint_flags_between_defaults_and_command_line():
    config = ConfigDict()
    config.algorithm = ConfigDict()
    config.algorithm.learning_rate, config.algorithm.nr_steps, config.algorithm.gamma = 3e-4, 128, 0.99
    stored = {"learning_rate": 1e-4, "nr_steps": 64, "unknown_flag": 7}
    adopt_checkpoint_config(config, stored, explicitly_set=["algorithm.nr_steps"])
    assert config.algorithm.learning_rate == 1e-4          # stored beats the default
    assert config.algorithm.nr_steps == 128                # the command line beats the stored value
    assert config.algorithm.gamma == 0.99 and "unknown_flag" not in config.algorithm
