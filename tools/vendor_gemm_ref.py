#!/usr/bin/env python
"""Reference point only (NOT used by the product): fp32 torch.matmul (rocBLAS / hipBLASLt) at the update's GEMM shapes."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
M = 32768
for name, (m, n, k, ta, tb) in {"fwd L2  [M,512]@[512,256]": (M, 256, 512, False, False), "fwd L3  [M,256]@[256,128]": (M, 128, 256, False, False),
                                "dx  L3  [M,128]@[128,256]": (M, 256, 128, False, True), "dx  L2  [M,256]@[256,512]": (M, 512, 256, False, True),
                                "dw  L2  [512,M]@[M,256]": (512, 256, M, True, False), "dw  L3  [256,M]@[M,128]": (256, 128, M, True, False)}.items():
    a = torch.randn((k, m) if ta else (m, k), device=dev)
    b = torch.randn((n, k) if tb else (k, n), device=dev)
    A = a.t() if ta else a
    B = b.t() if tb else b
    for _ in range(5):
        c = A @ B
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        c = A @ B
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name}: {us:8.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")
