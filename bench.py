#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: env-steps/sec (whole node) of the PPO hot path.

A "step" is ONE full PPO training iteration of configs[1]: synthetic random-obs env
(obs 17, act 6), 4096 envs PER GPU x 128 rollout steps (policy+critic inference, env step),
critic on next_states + GAE, bit-exact minibatch permutation, 10 epochs x 16 minibatch updates
(fused loss + MLP fwd/bwd on the matrix pipe + clip + Adam), then the metric reduction and its ONE device->host copy
(ppo.py::train_iteration -- exactly what the plugin's train() runs per loop turn).  Inputs are device-resident; weights are
random-init of the reference architecture (512-LN-256-128 ELU); data is synthetic.

Weak scaling over num_envs (BASELINE.json configs[2] / SURVEY.md 8(d) row 3: 32768 envs over 8 GPUs): every GPU keeps
4096 envs; the minibatch stays 32768 rows GLOBAL as the reference's default says, so an N-GPU iteration is
160 x N updates of ~32768 / N local rows each (launch- and collective-latency bound by construction).  With N > 1 the
same JSON line carries a clearly labelled secondary object `per_gpu_minibatch_variant`: 32768 rows PER GPU (global
minibatch 32768 x N, 160 updates at every N -- the usual data-parallel convention).  `--minibatch-size-global R`
overrides the headline run.  At N = 1 the line also carries `secondary_configs`: SAC at configs[3] and PPO+LSTM at
configs[4] shapes (short runs; `--no-secondary` skips them).

    python bench.py --gpus N --steps K --warmup W        (N > 1: launches one rank per GPU itself through torch.distributed.run,
                                                          or is started by it -- WORLD_SIZE set -- as the driver does)
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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_{f16,bf16}, dense (no sparsity)
BX_PRODUCTS = 3                 # fp16 MFMAs per fp32-equivalent product in the split-operand GEMMs (rl-x_amd/csrc/gemm_bx.h)
# The update's hidden-layer GEMMs run on the fp16 pipe with fp32 operands split into two fp16 planes; their algorithmic
# FLOPs stay 2 M N K (fp32 results to fp32 accuracy), so the peak they are priced against is the half-precision dense peak
# divided by the three plane products each fp32 product costs: 833.3 fp32-equivalent TFLOP/s (rounds 2-3: three bf16 planes,
# six products, 416.7).
BX_EQUIV_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / BX_PRODUCTS
ENVS_PER_GPU = 4096
NR_STEPS = 128
MINIBATCH_PER_GPU = 32768
PROF_SAMPLE = 25  # every 25th launch of each (kernel, engine, shape) row carries HIP events in the timed region (all: ~2 % slower)

# kernel actually launched for a (kind, engine) pair -- the names rocprofv3 prints (profiles/r04_bench_kernel_stats.md)
KERNEL_OF = {("k_gemm_fwd", 0): "k_gemm_fwd", ("k_gemm_fwd", 1): "k_gemm_bx<0,...>", ("k_gemm_dx", 0): "k_gemm_dx",
             ("k_gemm_dx", 1): "k_gemm_bx<1,...>", ("k_gemm_dw", 0): "k_gemm_dw", ("k_gemm_dw", 1): "k_gemm_dw_bx",
             ("k_dx_l1bwd", 0): "k_dx_l1bwd<..,false>", ("k_dx_l1bwd", 1): "k_dx_l1bwd<..,true>",
             ("k_l1fwd_mfma", 2): "k_l1fwd_mfma", ("k_head_loss", 2): "k_head_loss_fast", ("k_reduce_segments", 2): "k_reduce_segments",
             ("k_tail", 1): "k_tail_bx", ("k_l12fwd", 1): "k_l12fwd"}
ENGINE_HBM = 2            # profiler rows of the memory-bound kernels: priced against HBM bandwidth, algorithmic bytes / duration
ALGORITHMIC_BYTES_PER_ENV_STEP = 4300.0   # rollout write + read, minibatch gathers, params / Adam
HBM_PEAK_GBPS = 8000.0    # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def kernel_table(rows):
    """Per (kernel, engine, shape) row: every launch counted, 1 in PROF_SAMPLE timed (per-row counter).  Population figures
    are the timed means scaled by the launch counts; a kernel's total = sum over its shapes."""
    out = []
    for r in rows:
        if not r["timed"]:
            continue
        avg_ms = r["ms"] / r["timed"]
        if r["engine"] == ENGINE_HBM:
            gbps = (r["bytes"] / r["timed"]) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            out.append({"kernel": KERNEL_OF.get((r["kernel"], r["engine"]), r["kernel"]), "kind": r["kernel"], "engine": r["engine"],
                        "bound": "hbm", "shape_rows_width": [r["M"], r["N"], r["K"]], "launches": r["launches"],
                        "launches_timed": r["timed"], "avg_launch_us": round(1e3 * avg_ms, 2),
                        "total_ms_est": round(avg_ms * r["launches"], 3), "algorithmic_flops_per_launch": 0,
                        "algorithmic_bytes_per_launch": round(r["bytes"] / r["timed"]), "tflops": 0.0, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "algorithmic_GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4)})
            continue
        peak = BX_EQUIV_PEAK_TFLOPS if r["engine"] else F32_MFMA_PEAK_TFLOPS
        tf = (r["flops"] / r["timed"]) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        out.append({"kernel": KERNEL_OF.get((r["kernel"], r["engine"]), r["kernel"]), "kind": r["kernel"], "engine": r["engine"],
                    "shape_MNK": [r["M"], r["N"], r["K"]], "launches": r["launches"], "launches_timed": r["timed"],
                    "avg_launch_us": round(1e3 * avg_ms, 2), "total_ms_est": round(avg_ms * r["launches"], 3),
                    "algorithmic_flops_per_launch": round(r["flops"] / r["timed"]),
                    "algorithmic_bytes_per_launch": round(r["bytes"] / r["timed"]),
                    "tflops": round(tf, 2), "peak": round(peak, 1), "frac": round(tf / peak, 4),
                    "algorithmic_GBps": round((r["bytes"] / r["timed"]) / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else 0.0})
    out.sort(key=lambda x: -x["total_ms_est"])
    return out


def kernel_totals(table):
    """kernel name -> {total_ms_est, launches, flops, tflops, frac}: all shapes of a kernel together."""
    tot = {}
    for x in table:
        t = tot.setdefault(x["kernel"], {"total_ms_est": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "peak": x["peak"]})
        t["total_ms_est"] += x["total_ms_est"]
        t["launches"] += x["launches"]
        t["flops"] += x["algorithmic_flops_per_launch"] * x["launches"]
        t["bytes"] += x["algorithmic_bytes_per_launch"] * x["launches"]
    for t in tot.values():
        t["tflops"] = round(t["flops"] / max(t["total_ms_est"], 1e-9) / 1e9, 2)
        t["frac"] = round(t["tflops"] / t["peak"], 4)
        t["avg_launch_us"] = round(1e3 * t["total_ms_est"] / max(t["launches"], 1), 2)
        t["total_ms_est"] = round(t["total_ms_est"], 3)
    return tot


def pmc_traffic(kernel, table):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (tools/pmc_traffic.sh -> profiles/): a PMC
    pass cannot run inside this process.  The file holds one entry per (kernel, grid size), corrected as
    MI355X_MICROARCH.md prescribes (FETCH_SIZE KB x 1024 x 2 on gfx950, + WRITE_SIZE KB x 1024); a kernel's figure is
    the launch-weighted mean over its grids (= over its shapes)."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(tpath):
            continue
        try:
            tj = json.load(open(tpath))
        except Exception:
            continue
        kind = next((x["kind"] for x in table if x["kernel"] == kernel), kernel)
        for key in (kernel, kind):
            if key in tj.get("kernels", {}):
                return tj["kernels"][key]["hbm_bytes_per_launch"], os.path.relpath(tpath, ROOT), tj.get("rows", {}).get(key)
    return None, None, None



def pmc_update_traffic(updates_per_step, ms_per_step):
    """HBM bytes ONE minibatch update (both networks) moves, from the same committed PMC passes: sum over the update's kernels
    of bytes per launch x launches, divided by the updates of the pass (= launches of k_l12fwd / 2, one per network and update; the
    once-per-iteration k_pack_rows is spread over them).  The rate is the whole-iteration average (rollout and GAE time included),
    against the 8 TB/s HBM peak."""
    upd = ("k_gemm_dw_bx", "k_gemm_bx<0,...>", "k_gemm_bx<1,...>", "k_reduce_segments", "k_dx_l1bwd<..,true>", "k_head_loss_fast",
           "k_tail_bx", "k_tail32_bx", "k_l12fwd", "k_gather", "k_gather_rec", "k_pack_rows", "k_clip_adam", "k_l1fwd_mfma")
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        try:
            k = json.load(open(tpath))["kernels"]
            n_upd = k["k_l12fwd"]["launches"] // 2 if "k_l12fwd" in k else k["k_gather"]["launches"]
            total = sum(k[x]["hbm_bytes_per_launch"] * k[x]["launches"] for x in upd if x in k)
        except Exception:
            continue
        per_update = total / max(n_upd, 1)
        tbs = per_update * updates_per_step / (ms_per_step * 1e-3) / 1e12
        return {"note": "PMC HBM bytes of the update's kernels per minibatch update (both networks); rate = bytes per update x "
                        "updates per iteration / the measured iteration time (rollout and GAE included), peak 8 TB/s",
                "source": os.path.relpath(tpath, ROOT), "kernels": [x for x in upd if x in k],
                "bytes_per_update": round(per_update), "achieved_TBps": round(tbs, 3), "peak_TBps": 8.0, "frac": round(tbs / 8.0, 4),
                # ~4.3 kB algorithmic per env-step -> per update of a 160-update, 524288-step iteration
                "algorithmic_bytes_per_update": round(ALGORITHMIC_BYTES_PER_ENV_STEP * 524288 / 160),
                "counter_to_algorithmic_ratio": round(per_update / (ALGORITHMIC_BYTES_PER_ENV_STEP * 524288 / 160), 1)}
    return None


def _plugin(alg, env_overrides, alg_overrides):
    from rlx_amd.runner.config_dict import ConfigDict
    from rlx_amd.runner.default_config import get_config as runner_cfg
    from rlx_amd.algorithms.algorithm_manager import get_algorithm_config, get_algorithm_model_class
    from rlx_amd.environments.environment_manager import (get_environment_config,
                                                          get_environment_create_train_and_eval_env)
    config = ConfigDict()
    config.runner = runner_cfg("train")
    config.algorithm = get_algorithm_config(alg)
    config.environment = get_environment_config("synthetic.random_obs")
    for k, v in env_overrides.items():
        config.environment[k] = v
    for k, v in alg_overrides.items():
        config.algorithm[k] = v
    env, _ = get_environment_create_train_and_eval_env("synthetic.random_obs")(config)
    return get_algorithm_model_class(alg)(config, env, env, "/tmp/rlx_bench", None), env


def secondary_configs(torch):
    """BASELINE.json configs[3] (SAC) and configs[4] (PPO+LSTM) at their full shapes, short runs.  Roofline fractions
    use SURVEY.md 8(d)'s algorithmic FLOPs per unit against the pipe their GEMMs actually run on: the hidden-layer GEMMs of
    both are k_gemm_bx / k_gemm_dw_bx launches (split-fp32 operands on the fp16 pipe, 833.3 fp32-equivalent TFLOP/s);
    `frac_of_f32_mfma_peak` is the same rate against the 157.3 TFLOP/s exact-fp32 MFMA peak."""
    import rlx_amd.algorithms.sac.hip, rlx_amd.algorithms.ppo_lstm.hip  # noqa: F401,E401
    out = {}
    # ---- SAC: obs 376, act 17, replay 1M transitions, batch 4096, 4096 envs; one update per vector step
    for arch, gflop in (("flax", 21.9), ("full_jit", 47.7)):
        try:
            m, env = _plugin("sac.hip", dict(nr_envs=4096, obs_dim=376, act_dim=17),
                             dict(batch_size=4096, buffer_size=1_000_000, network_architecture=arch))
        except Exception as e:
            out[f"sac_configs3_{arch}"] = {"error": repr(e)}
            continue
        m._alloc()
        state, _ = env.reset()
        state = state.clone()

        def vector_step(state):          # the plugin's own per-step code (sac.py::train): act -> env.step -> replay add, then one update
            state = m.vector_step(env, state)
            m.sample_and_update()
            return state
        for _ in range(20):
            state = vector_step(state)
        torch.cuda.synchronize()
        K = 200
        t0 = time.perf_counter()
        for _ in range(K):
            state = vector_step(state)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ups = K / dt
        out[f"sac_configs3_{arch}"] = {
            "workload": f"SAC vector step (act + env + replay add + sample + update), obs 376 act 17, replay 1M, batch 4096, "
                        f"4096 envs, nets {'256-256 ReLU' if arch == 'flax' else '512-LN-256-128 ELU'}",
            "value": round(ups, 1), "unit": "updates/s", "env_steps_per_s": round(ups * 4096, 1),
            "ms_per_update": round(1e3 / ups, 3),
            "roofline": {"bound": "mfma", "algorithmic_GFLOP_per_update": gflop, "achieved": round(ups * gflop / 1e3, 2),
                         "peak": round(BX_EQUIV_PEAK_TFLOPS, 1), "unit": "TFLOP/s",
                         "frac": round(ups * gflop / 1e3 / BX_EQUIV_PEAK_TFLOPS, 4),
                         "frac_of_f32_mfma_peak": round(ups * gflop / 1e3 / F32_MFMA_PEAK_TFLOPS, 4)},
            "finite": bool(torch.isfinite(m.metrics_dev).all().item())}
        del m, env
    # ---- FastSAC (not a BASELINE config): the reference's default sizes -- batch 8192, 2 policy x 4 critic updates per
    #      vector step, 101 atoms, normaliser on -- obs 48 / act 12 (assumed), 4096 envs
    try:
        import rlx_amd.algorithms.fastsac.hip  # noqa: F401
        m, env = _plugin("fastsac.hip", dict(nr_envs=4096, obs_dim=48, act_dim=12), dict(buffer_size_per_env=64))
        m._alloc()
        state, _ = env.reset()
        state = state.clone()

        def fs_step(state, k):           # the plugin's own per-step code (fastsac.py::train)
            action = m.act(state)
            ns, r, term, trunc, _info = env.step(action)
            m.replay_add(state, ns, action, r, (term | trunc).float(), trunc.float())
            m.optimize(k)
            return ns.clone()
        for k in range(4):
            state = fs_step(state, k)
        torch.cuda.synchronize()
        K = 20
        t0 = time.perf_counter()
        for k in range(K):
            state = fs_step(state, k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["fastsac_default"] = {
            "workload": "FastSAC vector step (act + env + ring add + 65536-row sample + normaliser + 8 critic / 2 policy updates of 8192 rows), "
                        "obs 48 act 12, 4096 envs, nets 512-256-128 / 768-384-192 LayerNorm + SiLU, 101 atoms",
            "value": round(8 * K / dt, 1), "unit": "critic updates/s", "ms_per_vector_step": round(1e3 * dt / K, 3),
            "env_steps_per_s": round(K * 4096 / dt, 1), "finite": bool(torch.isfinite(m.metrics_c).all().item())}
        # algorithmic multiply-adds per row (2 FLOP each): a critic update = policy(s') + 2 target critics + 2 online critics forward
        # and backward (weight + input gradients, no input gradient below the first layer); a policy update = policy forward and
        # backward + both critics forward and input-gradient backward (fastsac/pytorch/fastsac.py:157-232)
        pol = (48, 512, 256, 128, 24)
        cri = (60, 768, 384, 192, 101)
        fwd = lambda d: sum(a * b for a, b in zip(d[:-1], d[1:]))
        bwd = lambda d: 2 * fwd(d) - d[0] * d[1]
        mac_c = fwd(pol) + 2 * fwd(cri) + 2 * (fwd(cri) + bwd(cri))
        mac_p = fwd(pol) + bwd(pol) + 2 * 2 * fwd(cri)
        gflop = 2.0 * (8192 * (8 * mac_c + 2 * mac_p) + 4096 * fwd(pol)) / 1e9
        tf = gflop / 1e3 / (dt / K)
        out["fastsac_default"]["roofline"] = {
            "bound": "mfma", "algorithmic_GFLOP_per_vector_step": round(gflop, 1), "achieved": round(tf, 2),
            "peak": round(BX_EQUIV_PEAK_TFLOPS, 1), "unit": "TFLOP/s", "frac": round(tf / BX_EQUIV_PEAK_TFLOPS, 4),
            "frac_of_f32_mfma_peak": round(tf / F32_MFMA_PEAK_TFLOPS, 4)}
        del m, env
    except Exception as e:
        out["fastsac_default"] = {"error": repr(e)}
    # ---- PPO+LSTM: 2048 envs x 128 steps, minibatch 32768 = 256 envs x 128 steps, 10 epochs
    try:
        m, env = _plugin("ppo_lstm.hip", dict(nr_envs=2048), dict(evaluation_and_save_frequency=-1))
        batch = m._alloc_batch()
        met = torch.zeros(m.nr_epochs * m.nr_minibatches, 10, device=m.device)
        state, _ = env.reset()
        state = state.contiguous()

        def iteration(state):
            state = m.collect_rollout(batch, state)
            m.compute_advantages(batch)
            m.update(batch, met)
            return state
        state = iteration(state)
        torch.cuda.synchronize()
        K = 4
        t0 = time.perf_counter()
        for _ in range(K):
            state = iteration(state)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        sps = 2048 * m.nr_steps / dt
        out["ppo_lstm_configs4"] = {
            "workload": "PPO+LSTM full training iteration, 2048 envs x 128 steps, obs 17 act 6 (assumed, SURVEY F9), H=64, "
                        "minibatch 32768 = 256 envs x 128 steps, 10 epochs (80 updates)",
            "value": round(sps, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * dt, 3),
            "roofline": {"bound": "mfma (with a serial-latency floor of 2 x 128 dependent cell steps per minibatch)",
                         "algorithmic_MFLOP_per_env_step": 30.67, "achieved": round(sps * 30.67e6 / 1e12, 2),
                         "peak": round(BX_EQUIV_PEAK_TFLOPS, 1), "unit": "TFLOP/s",
                         "frac": round(sps * 30.67e6 / 1e12 / BX_EQUIV_PEAK_TFLOPS, 4),
                         "frac_of_f32_mfma_peak": round(sps * 30.67e6 / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)},
            "finite": bool(torch.isfinite(met).all().item())}
    except Exception as e:
        out["ppo_lstm_configs4"] = {"error": repr(e)}
    return out


def self_launch_command(n_gpus, argv, port=None):
    """The command `python bench.py --gpus N` re-executes itself as (N > 1, WORLD_SIZE unset): the driver's own multi-GPU
    command line, on a free local port."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus):
    import subprocess
    cmd = self_launch_command(n_gpus, sys.argv[1:])
    if os.environ.get("RLX_BENCH_DRY_LAUNCH") == "1":      # tests: show the command instead of running it
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this driver (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


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
                    help="global minibatch rows of the headline run (default: 32768, the reference's default)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary runs (per-GPU-minibatch variant at N > 1; "
                                                                "SAC / PPO+LSTM configs at N = 1)")
    ap.add_argument("--no-prof", action="store_true", help="diagnostic: no per-kernel HIP events in the timed region")
    ap.add_argument("--lib-option", action="append", default=[], metavar="NAME=VALUE",
                    help="diagnostic: rlx_dbg_set_option before the run (e.g. two_streams=0)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as `python bench.py --gpus N` (the way the driver starts N = 1): become the launcher -- one rank per GPU under
        # torch.distributed.run on this node, rendezvous on 127.0.0.1, rank 0 prints the single JSON line
        sys.exit(self_launch(args.gpus))
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (python bench.py --gpus N launches them itself)")
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
    config.algorithm.minibatch_size = args.minibatch_size_global or MINIBATCH_PER_GPU
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

    def timed(steps, warmup, prof):
        nonlocal state
        for _ in range(warmup):
            state = model.train_iteration(batch, state, metrics)
        sync()
        if prof:
            model.ctx.prof_begin()                   # HIP events around the MFMA GEMM launches, on their stream
        t0 = time.perf_counter()
        for _ in range(steps):
            state = model.train_iteration(batch, state, metrics)
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=model.device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = tt.item()
        return dt

    # the same number of event pairs per iteration whatever the number of updates: at 1280 updates of 4096 rows (two free-running
    # chains of 12-20 us kernels) one launch in 25 with events costs 4 % (121.4 vs 116.6 ms per iteration); at the default 160
    # updates nothing (68.57 vs 68.6 ms)
    prof_sample = PROF_SAMPLE * max(1, (model.nr_epochs * model.nr_minibatches) // 160)
    model.ctx.set_option("prof_sample", prof_sample)
    for _ in range(args.warmup):                      # warm-up outside `timed` so that the collective counter brackets the K steps only
        state = model.train_iteration(batch, state, metrics)
    ar0 = model.ctx.get_counter("allreduce_calls")
    elapsed = timed(args.steps, 0, not args.no_prof)
    collectives_per_step = (model.ctx.get_counter("allreduce_calls") - ar0) / max(args.steps, 1)
    if args.no_prof:
        print(json.dumps({"value": args.steps * NR_STEPS * config.environment.nr_envs / elapsed,
                          "ms_per_step": 1e3 * elapsed / args.steps}))
        return
    model.ctx.prof_end()
    table_all = kernel_table(model.ctx.prof_rows())
    table = [x for x in table_all if x["engine"] != ENGINE_HBM]          # the matrix kernels
    hbm_rows = [x for x in table_all if x["engine"] == ENGINE_HBM]       # the memory-bound kernels, priced against HBM
    totals = kernel_totals(table)
    model.ctx.set_option("prof_sample", 1)
    model.check_distributed_health()
    # chip-level view: one extra UNTIMED iteration in which every launch carries events (union of the launch intervals)
    model.ctx.prof_begin()
    t1 = time.perf_counter()
    state = model.train_iteration(batch, state, metrics)
    prof_all = model.ctx.prof_end()
    full_iter_s = time.perf_counter() - t1
    union_ms = model.ctx.prof_union_ms()
    # kernel quality in isolation: one extra UNTIMED iteration with policy and critic serialised on one stream
    # (in the timed region they run concurrently on two streams, so per-launch durations overlap)
    fused_single = world == 1 and not args.force_distributed_update
    iso_totals = None
    if fused_single:
        model.ctx.set_option("two_streams", 0)
        model.ctx.prof_begin()
        state = model.train_iteration(batch, state, metrics)
        model.ctx.prof_end()
        iso_all = kernel_table(model.ctx.prof_rows())
        iso_table = [x for x in iso_all if x["engine"] != ENGINE_HBM]
        iso_hbm = [x for x in iso_all if x["engine"] == ENGINE_HBM]
        iso_totals = kernel_totals(iso_table)
        model.ctx.set_option("two_streams", 1)
    env_steps = args.steps * NR_STEPS * config.environment.nr_envs
    value = env_steps / elapsed
    finite = bool(torch.isfinite(metrics).all().item()) and bool(torch.isfinite(model.pparams).all().item())

    # roofline of the dominant kernel = the MFMA kernel with the largest TOTAL time over ALL its launches in the timed region
    dom = max(totals, key=lambda k: totals[k]["total_ms_est"])
    d = totals[dom]
    traffic, traffic_src, traffic_rows = pmc_traffic(dom, table)
    bx_on = os.environ.get("RLX_GEMM_BX", "1") != "0"
    # Which roof the dominant kernel sits nearer to is read off the measurement, not declared: its algorithmic FLOPs per launch
    # against the matrix-pipe peak and its algorithmic bytes per launch against the 8 TB/s HBM peak, both over the same
    # HIP-event launch duration; `bound` names the larger fraction and achieved / peak / unit / frac are that roof's figures
    # (the other roof's are kept beside it).  Neither fraction near 1 = the kernel is at neither roof (occupancy / phase
    # serialisation / co-scheduling): said in `bound_note`, with the update's whole-pass HBM view in `update_hbm`.
    alg_bytes = d["bytes"] / max(d["launches"], 1)
    dur_s = max(d["avg_launch_us"], 1e-9) * 1e-6
    hbm_gbs = alg_bytes / dur_s / 1e9
    view_mfma = {"achieved": d["tflops"], "peak": d["peak"], "unit": "TFLOP/s", "frac": d["frac"]}
    view_hbm = {"achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBPS, 4),
                "counter_bytes_frac": None if not traffic else round(traffic / dur_s / 1e9 / HBM_PEAK_GBPS, 4)}
    bound = "hbm" if view_hbm["frac"] >= view_mfma["frac"] else "mfma"
    chosen = view_hbm if bound == "hbm" else view_mfma
    upd_hbm = pmc_update_traffic(n_upd, 1e3 * elapsed / args.steps) if (world == 1 and n_upd == 160) else None
    roofline = {"bound": bound, "kernel": dom, "achieved": chosen["achieved"], "peak": chosen["peak"], "unit": chosen["unit"],
                "frac": chosen["frac"], "traffic": traffic,
                "bound_note": ("bound = the roof with the larger measured fraction for this kernel (mfma %.3f, hbm %.3f of peak over the "
                               "co-scheduled launch duration); below ~0.5 on both the kernel is limited by neither roof but by the "
                               "un-overlapped phases inside its workgroups and the other chain's kernels sharing the chip (DESIGN.md "
                               "section 4.6: a plain streaming kernel of the same residency reaches 5.4-6.2 TB/s)"
                               % (view_mfma["frac"], view_hbm["frac"])),
                "mfma": view_mfma, "hbm": view_hbm,
                "engine": ("split-fp32 operands (2 fp16 planes, 3 products) on v_mfma_f32_32x32x16_f16, fp32 accumulation; "
                           "mfma.achieved = fp32-equivalent algorithmic 2MNK / duration; mfma.peak = 2500 TFLOP/s dense fp16 / 3 products")
                          if d["peak"] != F32_MFMA_PEAK_TFLOPS else "exact fp32 v_mfma_f32_32x32x2_f32",
                "frac_of_f32_mfma_peak": round(d["tflops"] / F32_MFMA_PEAK_TFLOPS, 4),
                "traffic_source": traffic_src, "traffic_rows": traffic_rows,
                "algorithmic_bytes_per_launch": round(d["bytes"] / max(d["launches"], 1)),
                "algorithmic_flops_per_launch": round(d["flops"] / max(d["launches"], 1)),
                "launches": int(d["launches"]), "avg_launch_us": d["avg_launch_us"], "total_ms": d["total_ms_est"],
                "launch_sampling": f"every {prof_sample}th launch of each (kernel, engine, shape) row carries events (per-row "
                                   "counter: cannot alias with the launch pattern); totals = timed mean x launch count",
                "clock": "HIP events stamped at kernel start / end (hipExtLaunchKernelGGL) on the launch stream, inside the "
                         "timed region; the policy and critic chains run on two streams, so a launch shares the chip with "
                         "the other chain's kernels (co-scheduled duration; `isolated` = the same kernels alone)",
                "concurrent_streams": 2,
                "per_shape": table,
                "per_kernel": totals,
                "hbm_bound_kernels": {"note": "memory-bound kernels of the update, algorithmic HBM bytes / launch duration against "
                                              "8 TB/s: co-scheduled (timed region) and isolated (nets serialised, every launch timed)",
                                      "co_scheduled": hbm_rows, "isolated": iso_hbm if iso_totals is not None else None},
                "update_hbm": upd_hbm,
                "chip": {"note": "one extra untimed iteration with events on EVERY launch -- all MFMA kernels of both streams: sum "
                                 "of algorithmic FLOPs / union of their launch intervals",
                         "busy_ms": round(union_ms, 2),
                         "tflops": round(sum(v[1] for v in prof_all.values()) / max(union_ms, 1e-9) / 1e9, 2),
                         "frac": round(sum(v[1] for v in prof_all.values()) / max(union_ms, 1e-9) / 1e9
                                       / (BX_EQUIV_PEAK_TFLOPS if bx_on else F32_MFMA_PEAK_TFLOPS), 4),
                         "mfma_busy_fraction_of_iteration": round(union_ms * 1e-3 / full_iter_s, 4)},
                "isolated": None if iso_totals is None else {
                    "note": "same kernels, one extra untimed iteration with the two nets serialised on one stream (every launch timed)",
                    "kernel": dom, **({k: iso_totals[dom][k] for k in ("tflops", "frac", "avg_launch_us")} if dom in iso_totals else {}),
                    "per_kernel": iso_totals}}

    out = {
        "metric": "env-steps/sec (whole node) PPO 4096 envs at 1/2/4/8 MI355X",
        "value": round(value, 1), "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "dtype_note": "fp32 parameters, activations, gradients and accumulation; the hidden-layer GEMM operands enter the matrix "
                      "pipe as two fp16 planes (22 significant bits, power-of-two scaled; three plane products per fp32 product): "
                      "fp64-referenced error not above the exact-fp32 engine's (tests/test_gpu_gemm.py), bench shape vs the "
                      "float64 oracle at 1e-5 (tests/test_gpu_bench_shapes.py)",
        "config": {"workload": "PPO full training iteration, synthetic random-obs env obs=17 act=6 (BASELINE.json configs[1]"
                               + ("" if world == 1 else "; N > 1: configs[2] = SURVEY.md 8(d) row 3") + "); 4096 envs/GPU x 128 "
                               f"steps, 10 epochs, minibatch {int(model.minibatch_size)} rows GLOBAL, nets "
                               + ("512-LN-256-128 ELU" if args.arch == "full_jit" else "256-256 tanh"),
                   "nr_envs_global": int(config.environment.nr_envs), "nr_steps": NR_STEPS,
                   "minibatch_size_global": int(model.minibatch_size),
                   "updates_per_step": n_upd, "parallelism": f"dp{world} over num_envs"},
        "finite": finite, "roofline": roofline,
    }
    try:
        rccl_ranks = model.ctx.comm_count()
    except Exception:
        rccl_ranks = None
    out["multi_gpu"] = {
        "world_size": world, "rccl_comm_ranks": rccl_ranks,        # ncclCommCount of the library-owned communicator (0: none)
        "backend": ("none (single rank)" if world == 1 else os.environ.get("RLX_DIST_BACKEND", "nccl")),
        # counted by the library (rlx_dbg_get_counter "allreduce_calls" around the timed steps): advantage sums + metrics once per
        # step, per update ONE all-reduce over [policy | critic] gradients in the twin-launch schedule, else one per network
        "collectives_per_step": int(collectives_per_step) if collectives_per_step == int(collectives_per_step) else collectives_per_step,
        "collectives_per_step_source": "library counter (dist_allreduce calls of this rank's context over the timed steps / steps)",
        "scaling_curve": "NOT MEASURED by this run: one value at n_gpus = %d; the driver derives efficiency from its own 1/2/4/8 runs" % world,
    }
    if world > 1 and not args.no_secondary and not args.minibatch_size_global:
        # secondary, clearly labelled: 32768 minibatch rows PER GPU (global minibatch 32768 x N, 160 updates at every N)
        model.minibatch_size = MINIBATCH_PER_GPU * world
        model.nr_minibatches = model.batch_size // model.minibatch_size
        n2 = model.nr_epochs * model.nr_minibatches
        metrics = torch.zeros(n2, 10, device=model.device)
        el2 = timed(args.steps, max(1, args.warmup // 2), False)
        model.check_distributed_health()
        out["per_gpu_minibatch_variant"] = {
            "note": "SECONDARY, not the headline: 32768 minibatch rows per GPU instead of 32768 global",
            "value": round(env_steps / el2, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * el2 / args.steps, 3),
            "minibatch_size_global": int(model.minibatch_size), "updates_per_step": n2,
            "finite": bool(torch.isfinite(metrics).all().item())}
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            out["secondary_configs"] = secondary_configs(torch)
        except Exception as e:                                  # never lose the headline line to a secondary run
            out["secondary_configs"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.cpu_ppo_torch import time_iteration
        cb = time_iteration(arch="B" if args.arch == "full_jit" else "A")
        out["cpu_baseline"] = {"value": round(cb["env_steps_per_s"], 1), "unit": "env-steps/s", "cores": cb["cores"],
                               "kind": "port", "sample": cb["sample"],
                               "phases_s": {k: round(v, 5) for k, v in cb["phases_s"].items()}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
