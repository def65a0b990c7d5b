#!/usr/bin/env python
"""Time one PPO minibatch fwd+bwd (mb = 32768, arch B) per MFMA-kernel group; optional ablation variants."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from rlx_amd.hip import Ctx, PpoHparams, mlp_desc
dev = torch.device("cuda:0"); ctx = Ctx(0)
O, A, B, mb = 17, 6, 524288, 32768
pd = mlp_desc(O, [512, 256, 128], A, 1, True, True); cd = mlp_desc(O, [512, 256, 128], 1, 1, True, False)
import ctypes
npar = ctx.lib.rlx_mlp_param_count(ctypes.byref(pd)); ncar = ctx.lib.rlx_mlp_param_count(ctypes.byref(cd))
P = torch.randn(npar, device=dev) * 0.05; C = torch.randn(ncar, device=dev) * 0.05
P[-A:] = 0
states = torch.randn(B, O, device=dev); actions = torch.randn(B, A, device=dev)
logp = torch.randn(B, device=dev) * 0.1 - 8; ret = torch.randn(B, device=dev); adv = torch.randn(B, device=dev)
idx = torch.randperm(B, device=dev)[:mb].to(torch.int32)
pg = torch.zeros(npar, device=dev); cg = torch.zeros(ncar, device=dev); met = torch.zeros(8, device=dev)
hp = PpoHparams(0.1, 0.0, 1.0, 5.0, 0.9, 0.999, 1e-8)
def run(n):
    for _ in range(n):
        ctx.ppo_minibatch_fwd_bwd(pd, P, pg, cd, C, cg, met, states, actions, logp, ret, adv, idx, hp)
for variant in [int(v) for v in (sys.argv[1:] or ["0"])]:   # 0 = fused first-layer backward, 1 = unfused path
    ctx.set_option("disable_l1fused", variant)
    run(3); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.prof_begin(); e0.record(); run(10); e1.record(); p = ctx.prof_end()
    print(f"variant {variant}: minibatch fwd+bwd {e0.elapsed_time(e1)*100:.1f} us; " +
          "; ".join(f"{k} {1e3*v[0]/10:.1f} us/update" for k, v in p.items()))
