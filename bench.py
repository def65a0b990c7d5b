#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: env-steps/sec (whole node) of the PPO hot path.

A "step" is ONE full PPO training iteration of configs[1]: synthetic random-obs env
(obs 17, act 6), 4096 envs PER GPU x 128 rollout steps (policy+critic inference, env step),
critic on next_states + GAE, bit-exact minibatch permutation, 10 epochs x 16 minibatch updates
(fused loss + fp32-MFMA MLP fwd/bwd + clip + Adam).  Inputs are device-resident; weights are
random-init of the reference architecture (512-LN-256-128 ELU); data is synthetic.

Weak scaling over num_envs (BASELINE.json configs[2]: 32768 envs over 8 GPUs): every GPU keeps 4096
envs and 32768 minibatch rows, i.e. the GLOBAL minibatch is 32768 x N and the number of optimizer
updates per iteration stays 160 (the data-parallel convention: per-GPU work fixed).
`--minibatch-size-global 32768` keeps the reference's literal default instead (then N x more, N x
smaller updates per iteration; see DESIGN.md "Multi-GPU").

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
ENVS_PER_GPU = 4096
NR_STEPS = 128
MINIBATCH_PER_GPU = 32768


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="full_jit", choices=["full_jit", "flax"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-distributed-update", action="store_true",
                    help="diagnostic: run the multi-GPU update protocol on one rank (no collectives)")
    ap.add_argument("--minibatch-size-global", type=int, default=0,
                    help="global minibatch rows (default: 32768 per GPU, i.e. 32768 * N)")
    ap.add_argument("--lib-option", action="append", default=[], metavar="NAME=VALUE",
                    help="diagnostic: rlx_dbg_set_option before the run (e.g. l1bwd_pipelined=0)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    local_rank = min(local_rank, torch.cuda.device_count() - 1)   # (tests run 2 ranks on one GPU over gloo)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("RLX_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from rlx_amd.runner.config_dict import ConfigDict
    from rlx_amd.runner.default_config import get_config as runner_cfg
    import rlx_amd.algorithms.ppo.hip  # noqa: F401  (registers ppo.hip)
    import rlx_amd.environments.synthetic.random_obs  # noqa: F401
    from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
    from rlx_amd.environments.environment_manager import (get_environment_config,
                                                          get_environment_create_train_and_eval_env)

    config = ConfigDict()
    config.runner = runner_cfg("train")
    config.algorithm = get_algorithm_config("ppo.hip")
    config.environment = get_environment_config("synthetic.random_obs")
    config.algorithm.network_architecture = args.arch
    config.environment.nr_envs = ENVS_PER_GPU * world          # weak scaling: 4096 envs per GPU
    config.algorithm.force_distributed_update = args.force_distributed_update
    config.algorithm.minibatch_size = args.minibatch_size_global or MINIBATCH_PER_GPU * world
    train_env, eval_env = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
    model = get_algorithm_model_class("ppo.hip")(config, train_env, eval_env, "/tmp/rlx_bench", None)

    for kv in args.lib_option:
        name, val = kv.split("=")
        model.ctx.set_option(name, int(val))
    batch = model._alloc_batch()
    n_upd = model.nr_epochs * model.nr_minibatches
    metrics = torch.zeros(n_upd, 10, device=model.device)
    state, _ = train_env.reset()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        state = model.train_iteration(batch, state, metrics)
    sync()
    model.ctx.prof_begin()                       # HIP events around the MFMA GEMM launches, on their stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        state = model.train_iteration(batch, state, metrics)
    sync()
    elapsed = time.perf_counter() - t0
    prof = model.ctx.prof_end()
    union_ms = model.ctx.prof_union_ms()
    # kernel quality in isolation: one extra UNTIMED iteration with policy and critic serialised on one stream
    # (in the timed region they run concurrently on two streams, so per-launch durations overlap)
    fused_single = world == 1 and not args.force_distributed_update
    prof_iso = None
    if fused_single:
        model.ctx.set_option("two_streams", 0)
        model.ctx.prof_begin()
        state = model.train_iteration(batch, state, metrics)
        prof_iso = model.ctx.prof_end()
        model.ctx.set_option("two_streams", 1)
    if world > 1:
        tt = torch.tensor([elapsed], device=model.device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()

    env_steps = args.steps * NR_STEPS * config.environment.nr_envs
    value = env_steps / elapsed
    finite = bool(torch.isfinite(metrics).all().item()) and bool(torch.isfinite(model.pparams).all().item())

    # roofline of the dominant kernel (largest total time among the MFMA kernels, timed live with HIP events)
    dom = max(prof, key=lambda k: prof[k][0])
    ms, flops, cnt, abytes = prof[dom]
    achieved = (flops / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0
    # HBM traffic per launch from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh -> profiles/): a PMC
    # pass cannot run inside this process; the file holds bytes per launch of the same command, corrected as
    # MI355X_MICROARCH.md prescribes (FETCH_SIZE KB x 1024 x 2 on gfx950, + WRITE_SIZE KB x 1024)
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if dom in tj.get("kernels", {}):
                traffic, traffic_src = tj["kernels"][dom]["hbm_bytes_per_launch"], "profiles/r01_pmc_traffic.json"
        except Exception:
            pass
    roofline = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(abytes / max(cnt, 1)),
                "algorithmic_flops_per_launch": round(flops / max(cnt, 1)),
                "launches": int(cnt), "avg_launch_us": round(1e3 * ms / max(cnt, 1), 2),
                "concurrent_streams": 2,
                "chip": {"note": "all MFMA kernels of both streams: sum of algorithmic FLOPs / union of their launch intervals",
                         "busy_ms": round(union_ms, 2),
                         "tflops": round(sum(v[1] for v in prof.values()) / max(union_ms, 1e-9) / 1e9, 2),
                         "frac": round(sum(v[1] for v in prof.values()) / max(union_ms, 1e-9) / 1e9 / F32_MFMA_PEAK_TFLOPS, 4)},
                "isolated": None if prof_iso is None else {
                    "note": "same kernels, one extra untimed iteration with the two nets serialised on one stream",
                    "kernel": dom, "tflops": round(prof_iso[dom][1] / max(prof_iso[dom][0], 1e-9) / 1e9, 2),
                    "frac": round(prof_iso[dom][1] / max(prof_iso[dom][0], 1e-9) / 1e9 / F32_MFMA_PEAK_TFLOPS, 4),
                    "avg_launch_us": round(1e3 * prof_iso[dom][0] / max(prof_iso[dom][2], 1), 2),
                    "all_mfma_kernels": {k: {"tflops": round(v[1] / max(v[0], 1e-9) / 1e9, 2),
                                             "avg_launch_us": round(1e3 * v[0] / max(v[2], 1), 2)}
                                         for k, v in prof_iso.items() if v[2]}},
                "all_mfma_kernels": {k: {"ms": round(v[0], 2), "tflops": round(v[1] / max(v[0], 1e-9) / 1e9, 2),
                                         "launches": int(v[2]),
                                         "algorithmic_GBps": round(v[3] / max(v[0], 1e-9) / 1e6, 1)}
                                     for k, v in prof.items() if v[2]},
                "mfma_busy_fraction_of_step": round(union_ms * 1e-3 / elapsed, 4)}

    out = {
        "metric": "env-steps/sec (whole node) PPO 4096 envs at 1/2/4/8 MI355X",
        "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "PPO full training iteration, synthetic random-obs env obs=17 act=6 "
                               "(BASELINE.json configs[1]); 4096 envs/GPU x 128 steps, 10 epochs, "
                               "32768 minibatch rows/GPU, nets " + ("512-LN-256-128 ELU" if args.arch == "full_jit"
                                                                     else "256-256 tanh"),
                   "nr_envs_global": int(config.environment.nr_envs), "nr_steps": NR_STEPS,
                   "minibatch_size_global": int(model.minibatch_size),
                   "updates_per_step": n_upd, "parallelism": f"dp{world} over num_envs"},
        "finite": finite, "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.cpu_ppo_torch import time_iteration
        cb = time_iteration(arch="B" if args.arch == "full_jit" else "A")
        out["cpu_baseline"] = {"value": round(cb["env_steps_per_s"], 1), "unit": "env-steps/s", "cores": cb["cores"],
                               "kind": "port", "sample": cb["sample"]}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
