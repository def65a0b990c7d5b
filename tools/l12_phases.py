#!/usr/bin/env python
"""Tuning aid: phase stamps (clock64 of thread 0, workgroup 0) of k_l12fwd at the bench minibatch, every kernel alone on the chip.
Needs a library built with the stamps compiled in: RLX_EXTRA_DEFINES=-DRLX_L12_STAMPS=1 python rl-x_amd/build.py --force"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, ROOT)
import torch
from rlx_amd.hip import Ctx, PpoHparams, mlp_desc
dev = torch.device("cuda:0")
ctx = Ctx(0)
O, A, B, mb = 17, 6, 524288, int(sys.argv[1]) if len(sys.argv) > 1 else 32768
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
for _ in range(3):
    ctx.ppo_minibatch_fwd_bwd(pd, P, pg, cd, C, cg, met, states, actions, logp, ret, adv, idx, hp)
st = torch.zeros(16, dtype=torch.int64, device=dev)
ctx.dbg_set_stamps(st)
ctx.ppo_minibatch_fwd_bwd(pd, P, pg, cd, C, cg, met, states, actions, logp, ret, adv, idx, hp)
torch.cuda.synchronize()
ctx.dbg_set_stamps(None)
s = st.cpu().numpy()
names = ["prologue", "z1 MFMA + next x stage", "LN sums + barrier", "normalise + ELU + h1 store + image + barrier", "layer-2 K loop", "h2 epilogue + store", "loop-top barrier"]
d = [int(s[i + 1] - s[i]) for i in range(12)]
wall_us = (int(s[15]) - int(s[14])) / 100.0
ticks = int(s[12] - s[0])
print(f"k_l12fwd workgroup 0 (the critic's launch), mb {mb}: two tiles = {ticks} clock64 ticks in {wall_us:.2f} us of wall_clock64 -> {ticks / wall_us / 1e3:.2f} GHz")
print("  tile 0: " + ", ".join(f"{n} {v}" for n, v in zip(names, d[:7])))
print("  tile 1: " + ", ".join(f"{n} {v}" for n, v in zip(names[1:6], d[7:12])))
