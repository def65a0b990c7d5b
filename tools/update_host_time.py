#!/usr/bin/env python
"""Host issue time vs device time of ONE rlx_ppo_update_f32 call (arch B, 4096 envs x 128 steps, 10 epochs):

    python tools/update_host_time.py [--mb 4096]

`host` = wall time of the call itself (it only enqueues), `total` = until the device is idle.  host ~= total: the update is
bound by the launch rate of the issuing thread, not by the kernels."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from rlx_amd.hip import Ctx, PpoHparams, mlp_desc  # noqa: E402
from rlx_amd.hip import lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mb", type=int, default=4096)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("settings", nargs="*", default=[""])
args = ap.parse_args()
dev = torch.device("cuda:0")
ctx = Ctx(0)
O, A, T, N, E = 17, 6, 128, 4096, 10
pd = mlp_desc(O, [512, 256, 128], A, 1, True, True)
cd = mlp_desc(O, [512, 256, 128], 1, 1, True, False)
npar, ncar = ctx.lib.rlx_mlp_param_count(ctypes.byref(pd)), ctx.lib.rlx_mlp_param_count(ctypes.byref(cd))
P, C = torch.randn(npar, device=dev) * 0.05, torch.randn(ncar, device=dev) * 0.05
P[-A:] = 0
z = torch.zeros_like
pm, pv, cm, cv = z(P), z(P), z(C), z(C)
S, Ac = torch.randn(T, N, O, device=dev), torch.randn(T, N, A, device=dev)
LP, R, AD = torch.randn(T, N, device=dev) * 0.1 - 8, torch.randn(T, N, device=dev), torch.randn(T, N, device=dev)
hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)
n_upd = E * (T * N // args.mb)
met = torch.empty(n_upd, 10, device=dev)
lr = np.full(n_upd, 1e-5, np.float32)
key, cnt = L.prng_key(1), 0
for setting in args.settings:
    for kv in [kv.split("=") for kv in setting.split(",") if kv]:
        ctx.set_option(kv[0], int(kv[1]))
    for rep in range(args.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        key, cnt = ctx.ppo_update(pd, P, pm, pv, cd, C, cm, cv, S, Ac, LP, R, AD, E, args.mb, key, cnt, lr, hp, met)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if rep:
            print(f"[{setting or 'default'}] mb {args.mb}: {n_upd} updates, host {1e3 * (t1 - t0):7.1f} ms, total {1e3 * (t2 - t0):7.1f} ms "
                  f"({1e6 * (t2 - t0) / n_upd:6.1f} us / update, host {1e6 * (t1 - t0) / n_upd:6.1f})")
