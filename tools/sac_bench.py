#!/usr/bin/env python
"""SAC at BASELINE configs[3] shapes: obs 376, act 17, replay 1M, batch 4096, 4096 envs (assumed).
Reports updates/s and env-steps/s (= nr_envs x vector steps/s; one update per vector step)."""
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
if "full_jit" in sys.argv:          # sac/flax_full_jit nets (512-LayerNorm-256-128 ELU), device-side index draws
    config.algorithm.network_architecture = "full_jit"
env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
m = get_algorithm_model_class("sac.hip")(config, env, env, "/tmp/x", None)
m._alloc()
state, _ = env.reset(); state = state.clone()

def vector_step(state):
    state = m.vector_step(env, state)      # the plugin's own per-step code
    m.sample_and_update()
    return state

opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)     # e.g. sac_twin=0 two_streams=0
for k, v in opts.items():
    m.ctx.set_option(k, int(v))
for _ in range(20): state = vector_step(state)
K = 300
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): state = vector_step(state)
    t_host = time.perf_counter() - t0            # the host has SUBMITTED everything (it does not wait for the GPU in the loop)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"SAC {opts}: {K/dt:.1f} updates/s, {K*4096/dt/1e6:.3f} M env-steps/s, {1e3*dt/K:.3f} ms per vector step "
          f"(host submission {1e3*t_host/K:.3f} ms per step: {'host' if t_host > 0.95 * dt else 'GPU'}-bound)")
if "prof" in sys.argv:
    m.ctx.prof_begin()
    for _ in range(50): state = vector_step(state)
    p = m.ctx.prof_end()
    print("GEMM kernels (eager, instrumented): " + ", ".join(f"{k}: {v[0]/50*1e3:.0f} us/update {v[1]/max(v[0],1e-9)/1e9:.1f} TF" for k, v in p.items() if v[2]))
print("finite:", bool(torch.isfinite(m.metrics_dev).all()), m.metrics_dev.cpu().tolist()[:6])
