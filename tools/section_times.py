#!/usr/bin/env python
"""Per-section wall time of one PPO iteration (sync between sections) -- a tuning aid."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import torch  # noqa: E402
from rlx_amd.runner.config_dict import ConfigDict  # noqa: E402
from rlx_amd.runner.default_config import get_config as runner_cfg  # noqa: E402
import rlx_amd.algorithms.ppo.hip  # noqa: E402,F401
import rlx_amd.environments.synthetic.random_obs  # noqa: E402,F401
from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class  # noqa: E402
from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env  # noqa: E402

config = ConfigDict()
config.runner = runner_cfg("train")
config.algorithm = get_algorithm_config("ppo.hip")
config.environment = get_environment_config("synthetic.random_obs")
env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
model = get_algorithm_model_class("ppo.hip")(config, env, env, "/tmp/x", None)
batch = model._alloc_batch()
met = torch.zeros(model.nr_epochs * model.nr_minibatches, 10, device=model.device)
state, _ = env.reset()
PREFETCH = os.environ.get("PREFETCH", "0") == "1"
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    state = model.collect_rollout(batch, state)
    if PREFETCH:
        model.prefetch_permutation(rollout_queued=True)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter()
    model.compute_advantages(batch)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    model.update(batch, met)
    t_issue_u = time.perf_counter() - t2
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"iter {it}: rollout {1e3*(t1-t0):.1f} ms (host issue {1e3*t_issue:.1f}), adv {1e3*(t2-t1):.1f} ms, "
          f"update {1e3*(t3-t2):.1f} ms (host issue {1e3*t_issue_u:.1f})")
