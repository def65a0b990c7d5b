#!/usr/bin/env python
"""SAC vector step at configs[3] shapes: who bounds it?  Per step, with the queue EMPTY at its start (a synchronize in front):
host time to submit the step, and time until the device has finished it (= the step's dependent chain incl. launch latencies).
Then the usual free-running rate.  `python tools/sac_host_time.py [full_jit] [option=value ...]`"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import torch
from rlx_amd.runner.config_dict import ConfigDict
from rlx_amd.runner.default_config import get_config as runner_cfg
import rlx_amd.algorithms.sac.hip, rlx_amd.environments.synthetic.random_obs  # noqa
from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env

config = ConfigDict()
config.runner = runner_cfg("train")
config.algorithm = get_algorithm_config("sac.hip")
config.environment = get_environment_config("synthetic.random_obs")
config.environment.nr_envs, config.environment.obs_dim, config.environment.act_dim = 4096, 376, 17
config.algorithm.batch_size, config.algorithm.buffer_size = 4096, 1_000_000
if "full_jit" in sys.argv:
    config.algorithm.network_architecture = "full_jit"
env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
m = get_algorithm_model_class("sac.hip")(config, env, env, "/tmp/x", None)
m._alloc()
for k, v in (a.split("=") for a in sys.argv[1:] if "=" in a):
    m.ctx.set_option(k, int(v))
state, _ = env.reset(); state = state.clone()

def step(state):
    state = m.vector_step(env, state)
    m.sample_and_update()
    return state

for _ in range(50): state = step(state)
torch.cuda.synchronize()
K = 300
hs = ds = 0.0
parts = [0.0, 0.0]
for _ in range(K):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    state = m.vector_step(env, state)
    ta = time.perf_counter()
    m.sample_and_update()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs += t1 - t0; ds += t2 - t0; parts[0] += ta - t0; parts[1] += t1 - ta
print(f"isolated step: host submission {1e6*hs/K:.1f} us (act+env {1e6*parts[0]/K:.1f}, sample+update {1e6*parts[1]/K:.1f}), "
      f"finished on the device after {1e6*ds/K:.1f} us")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): state = step(state)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"free running: {K/dt:.1f} updates/s = {1e6*dt/K:.1f} us per step (host loop {1e6*th/K:.1f} us)")
