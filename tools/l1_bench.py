#!/usr/bin/env python
"""Micro-benchmark of the fused first-layer kernel (fwd / bwd) at M = 32768, O = 17, H = 512."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import torch
from rlx_amd.hip import Ctx
dev = torch.device("cuda:0"); ctx = Ctx(0)
M, O, Hd = 32768, 17, int(sys.argv[1]) if len(sys.argv) > 1 else 512
X = torch.randn(M, O, device=dev); W = torch.randn(O, Hd, device=dev) * 0.3
b = torch.zeros(Hd, device=dev); g = torch.ones(Hd, device=dev); be = torch.zeros(Hd, device=dev)
H = torch.empty(M, Hd, device=dev)
for grid in (256, 512, 768, 1024):
    lnp = torch.empty(grid, 2 * Hd, device=dev)
    for bwd in (0, 1):
        for act, ln in ((1, 1), (0, 0), (2, 0)):
            for _ in range(3): ctx.dbg_l1(bwd, X, W, b, g, be, H, lnp, act, ln, grid)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ctx.dbg_l1(bwd, X, W, b, g, be, H, lnp, act, ln, grid)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            gb = M * Hd * 4 * (2 if bwd else 1) / 1e9
            print(f"grid {grid:5d} bwd={bwd} act={act} ln={ln}: {us:7.1f} us  {gb/us*1e6/1e3:6.2f} TB/s")
