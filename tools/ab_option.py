#!/usr/bin/env python
"""A/B a library option inside ONE process (box-to-box and run-to-run variance is several per cent, larger than most kernel
changes): alternates blocks of PPO bench iterations with the option at value A and at value B.
    python tools/ab_option.py gemm_bx 1 0 [--blocks 4] [--iters 5]
    python tools/ab_option.py attr:rollout_one_call 1 0        (an attribute of the plugin object instead of a library option)"""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("option")
ap.add_argument("a", type=int)
ap.add_argument("b", type=int)
ap.add_argument("--blocks", type=int, default=4)
ap.add_argument("--iters", type=int, default=5)
args = ap.parse_args()
config = ConfigDict()
config.runner = runner_cfg("train")
config.algorithm = get_algorithm_config("ppo.hip")
config.environment = get_environment_config("synthetic.random_obs")
env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
model = get_algorithm_model_class("ppo.hip")(config, env, env, "/tmp/x", None)
batch = model._alloc_batch()
met = torch.zeros(model.nr_epochs * model.nr_minibatches, 10, device=model.device)
state, _ = env.reset()
def set_option(name, v):
    if name.startswith("attr:"):
        setattr(model, name[5:], type(getattr(model, name[5:]))(v))
    else:
        model.ctx.set_option(name, v)


for v in (args.a, args.b):
    set_option(args.option, v)
    for _ in range(2):
        state = model.train_iteration(batch, state, met)
res = {args.a: [], args.b: []}
for blk in range(args.blocks):
    for v in (args.a, args.b) if blk % 2 == 0 else (args.b, args.a):
        set_option(args.option, v)
        state = model.train_iteration(batch, state, met)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            state = model.train_iteration(batch, state, met)
        torch.cuda.synchronize()
        res[v].append(1e3 * (time.perf_counter() - t0) / args.iters)
for v, xs in res.items():
    print(f"{args.option}={v}: " + " ".join(f"{x:.2f}" for x in xs) + f"  | mean {sum(xs)/len(xs):.2f} ms/iteration")
