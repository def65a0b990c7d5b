#!/usr/bin/env python
"""Tuning aid: T acting steps of the bench configuration with k_rollout_step returning after a phase (option ro_exit = 1 / 2 / 3:
after layer 0 / the hidden layers / the head; 0: the whole kernel).  Run under rocprofv3 by tools/rollout_phases.sh."""
import os
import sys

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
state, _ = env.reset()
model.ctx.set_option("ro_exit", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for _ in range(4):
    state = model.collect_rollout(batch, state)
torch.cuda.synchronize()
if len(sys.argv) > 2 and sys.argv[2] == "stamps":      # phase stamps of workgroup 0 (a policy workgroup), last step of one more rollout
    st = torch.zeros(16, dtype=torch.int64, device=model.device)
    model.ctx.dbg_set_stamps(st)
    state = model.collect_rollout(batch, state)
    torch.cuda.synchronize()
    model.ctx.dbg_set_stamps(None)
    s = st.cpu().numpy()
    names = ["start -> obs tile in LDS", "W0 regs + 36 MFMA", "LayerNorm sums + barrier", "normalise + ELU + stores + barrier", "layer 1",
             "layer 2", "head", "sample / log-prob + pre-reset observations", "per-row env", "reset observations + stores"]
    print("k_rollout_step, workgroup 0, clock64 ticks of 10 ns: " + ", ".join(f"{n} {int(s[i + 1] - s[i])}" for i, n in enumerate(names))
          + f"; total {int(s[10] - s[0])}")
