#!/usr/bin/env python
"""Time ONE PPO minibatch fwd+bwd of both nets (arch B, serial on one stream: every kernel alone on the chip), per
(kernel, engine, shape) row of the library's profiler, for a list of option settings:

    python tools/mb_bench.py [--mb 32768] [--reps 20] [name=value[,name=value...] ...]

e.g. `python tools/mb_bench.py gemm_bx=1 gemm_bx=0` prints one table per setting (all other options default)."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from rlx_amd.hip import Ctx, PpoHparams, mlp_desc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=int, default=32768)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("settings", nargs="*", default=[""])
args = ap.parse_args()
dev = torch.device("cuda:0")
ctx = Ctx(0)
O, A, B, mb = 17, 6, 524288, args.mb
pd = mlp_desc(O, [512, 256, 128], A, 1, True, True)
cd = mlp_desc(O, [512, 256, 128], 1, 1, True, False)
npar, ncar = ctx.lib.rlx_mlp_param_count(ctypes.byref(pd)), ctx.lib.rlx_mlp_param_count(ctypes.byref(cd))
P, C = torch.randn(npar, device=dev) * 0.05, torch.randn(ncar, device=dev) * 0.05
P[-A:] = 0
states, actions = torch.randn(B, O, device=dev), torch.randn(B, A, device=dev)
logp, ret, adv = torch.randn(B, device=dev) * 0.1 - 8, torch.randn(B, device=dev), torch.randn(B, device=dev)
idx = torch.randperm(B, device=dev)[:mb].to(torch.int32)
pg, cg, met = torch.zeros(npar, device=dev), torch.zeros(ncar, device=dev), torch.zeros(8, device=dev)
hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)


def run(n):
    for _ in range(n):
        ctx.ppo_minibatch_fwd_bwd(pd, P, pg, cd, C, cg, met, states, actions, logp, ret, adv, idx, hp)


ref = None
for setting in args.settings:
    opts = [kv.split("=") for kv in setting.split(",") if kv]
    for k, v in opts:
        ctx.set_option(k, int(v))
    run(3)
    torch.cuda.synchronize()
    g = (pg.clone(), cg.clone())
    if ref is None:
        ref = g
    dev_rel = [((a - b).norm() / b.norm()).item() for a, b in zip(g, ref)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.prof_begin()
    e0.record()
    run(args.reps)
    e1.record()
    ctx.prof_end()
    rows = ctx.prof_rows()
    print(f"== {setting or 'defaults'}: minibatch fwd+bwd (both nets, one stream) {e0.elapsed_time(e1) * 1e3 / args.reps:.1f} us; "
          f"gradients vs the first setting: policy {dev_rel[0]:.1e} critic {dev_rel[1]:.1e}")
    tot = 0.0
    for r in sorted(rows, key=lambda r: -r["ms"]):
        us = 1e3 * r["ms"] / max(r["timed"], 1)
        tot += 1e3 * r["ms"] / args.reps
        tf = r["flops"] / max(r["ms"], 1e-9) / 1e9
        gbs = r["bytes"] / max(r["ms"], 1e-9) / 1e6
        print(f"   {r['kernel']:18s} engine {r['engine']} MNK {r['M']:6d} {r['N']:4d} {r['K']:6d}: {us:7.1f} us/launch x "
              f"{r['launches'] // args.reps} = {1e3 * r['ms'] / args.reps:7.1f} us/update  {tf:6.1f} TF  {gbs:7.0f} GB/s (algorithmic)")
    print(f"   instrumented kernels {tot:.1f} us/update")
    for k, v in opts:      # back to the defaults the library documents
        ctx.set_option(k, {"gemm_bx": 1, "two_streams": 1, "adam_emit": 1, "l1fwd_mfma": 1, "sac_twin": 1, "pipeline_updates": 1}.get(k, 0))
