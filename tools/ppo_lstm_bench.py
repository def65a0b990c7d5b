#!/usr/bin/env python
"""PPO+LSTM at BASELINE configs[4] shapes: 2048 envs x 128 steps, obs 17 / act 6 (assumed), H=64,
minibatch 32768 (= 256 envs x 128 steps), 10 epochs.  Reports env-steps/s and the phase split."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import torch
from rlx_amd.runner.config_dict import ConfigDict
from rlx_amd.runner.default_config import get_config as runner_cfg
import rlx_amd.algorithms.ppo_lstm.hip, rlx_amd.algorithms.ppo_gru.hip, rlx_amd.environments.synthetic.random_obs  # noqa
from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
from rlx_amd.environments.environment_manager import get_environment_config, get_environment_create_train_and_eval_env

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ALG = sys.argv[2] if len(sys.argv) > 2 else "ppo_lstm.hip"      # or ppo_gru.hip
config = ConfigDict()
config.runner = runner_cfg("train")
config.algorithm = get_algorithm_config(ALG)
config.environment = get_environment_config("synthetic.random_obs")
config.environment.nr_envs = N
config.algorithm.evaluation_and_save_frequency = -1
env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
m = get_algorithm_model_class(ALG)(config, env, env, "/tmp/x", None)
for kv in [x for x in os.environ.get("RLX_OPTS", "").split(",") if x]:      # e.g. RLX_OPTS=gemm_bx=0
    m.ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
batch = m._alloc_batch()
met = torch.zeros(m.nr_epochs * m.nr_minibatches, 10, device=m.device)
state, _ = env.reset()
state = state.contiguous()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
K = 3
for it in range(K + 1):
    if it == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter(); acc = [0.0, 0.0, 0.0]
    ev[0].record(); state = m.collect_rollout(batch, state)
    ev[1].record(); m.compute_advantages(batch)
    ev[2].record(); m.update(batch, met)
    ev[3].record(); torch.cuda.synchronize()
    if it >= 1:
        for i in range(3):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
dt = (time.perf_counter() - t0) / K
print(f"{ALG} N={N} T={m.nr_steps}: {1e3*dt:.1f} ms/iteration, {N*m.nr_steps/dt/1e6:.3f} M env-steps/s; "
      f"rollout {acc[0]/K:.1f} ms, gae {acc[1]/K:.1f} ms, update {acc[2]/K:.1f} ms ({acc[2]/K/met.shape[0]:.2f} ms/minibatch)")
print("finite:", bool(torch.isfinite(met).all()), met.mean(0).cpu().tolist())
