#!/usr/bin/env python
"""Micro-benchmark of the three MFMA GEMM kernels at the PPO update shapes (M = 32768)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import torch  # noqa: E402
from rlx_amd.hip import Ctx  # noqa: E402

dev = torch.device("cuda:0")
ctx = Ctx(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
if len(sys.argv) > 2:
    ctx.set_option("bx_force_mi", int(sys.argv[2]))
shapes = {0: [(256, 512), (128, 256)], 1: [(256, 512), (128, 256)], 2: [(256, 512), (128, 256)]}
names = ("k_gemm_fwd", "k_gemm_dx", "k_gemm_dw")
for mode in (0, 1, 2, 3, 4, 5):      # 3-5: the same kernels on the fp16 pipe with split-fp32 operands
    bx, mode = (3, mode - 3) if mode >= 3 else (0, mode)
    for (N, K) in shapes[mode]:
        if mode == 0:
            A, B, C, aux = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev) * 0.05, torch.empty(M, N, device=dev), torch.zeros(N, device=dev)
        elif mode == 1:
            A, B, C, aux = torch.randn(M, N, device=dev), torch.randn(K, N, device=dev) * 0.05, torch.randn(M, K, device=dev), None
        else:
            A, B, C, aux = torch.randn(M, K, device=dev), torch.randn(M, N, device=dev), torch.empty(K, N, device=dev), torch.zeros(N, device=dev)
        for _ in range(3):
            ctx.dbg_gemm(mode + bx, A, B, C, aux, M, N, K, 1)
        torch.cuda.synchronize()
        ctx.prof_begin()
        for _ in range(20):
            ctx.dbg_gemm(mode + bx, A, B, C, aux, M, N, K, 1)
        p = ctx.prof_end()[names[mode]]
        print(f"{names[mode] + ('_bx' if bx else ''):14s} M={M} N={N} K={K}: {1e3*p[0]/p[2]:8.1f} us  {p[1]/p[0]/1e9:7.1f} TFLOP/s")
