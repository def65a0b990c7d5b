#!/usr/bin/env python
"""`fastsac.hip` vector step at the reference's default sizes (fastsac/pytorch/default_config.py: batch 8192, 2 policy x 4 critic
updates per step, nr_atoms 101) on the synthetic env: obs 48 / act 12 (assumed), 4096 envs.  Reports vector steps/s, critic
updates/s and env-steps/s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import torch
from rlx_amd.runner.config_dict import ConfigDict
from rlx_amd.runner.default_config import get_config as runner_cfg
import rlx_amd.algorithms.fastsac.hip, rlx_amd.environments.synthetic.random_obs  # noqa
from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env

config = ConfigDict()
config.runner = runner_cfg("train")
config.algorithm = get_algorithm_config("fastsac.hip")
config.environment = get_environment_config("synthetic.random_obs")
config.environment.nr_envs, config.environment.obs_dim, config.environment.act_dim = 4096, 48, 12
config.algorithm.buffer_size_per_env = 64
env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
m = get_algorithm_model_class("fastsac.hip")(config, env, env, "/tmp/x", None)
m._alloc()
state, _ = env.reset(); state = state.clone()

def vector_step(state, k):
    action = m.act(state)
    ns, r, term, trunc, info = env.step(action)
    m.replay_add(state, ns, action, r, (term | trunc).float(), trunc.float())
    m.optimize(k)
    return ns.clone()

for k in range(4): state = vector_step(state, k)
K = 20
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K): state = vector_step(state, k)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"FastSAC: {1e3*dt/K:.2f} ms per vector step (host {1e3*t_host/K:.2f}), {8*K/dt:.0f} critic updates/s, {K*4096/dt/1e6:.3f} M env-steps/s")
print("finite:", bool(torch.isfinite(m.metrics_c).all()), m.metrics_c.cpu().tolist())
