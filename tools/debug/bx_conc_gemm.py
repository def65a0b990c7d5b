"""Repeat each GEMM kernel of both engines in two contexts on two streams at once; any bitwise run-to-run difference is a race."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import torch
from rlx_amd.hip import Ctx
dev = torch.device("cuda:0")
ctxs = (Ctx(0), Ctx(0))
streams = (torch.cuda.current_stream(), torch.cuda.Stream())
M = 32768
for mode in (0, 1, 2, 3, 4, 5):
    for (N, K) in ((256, 512), (128, 256)):
        torch.manual_seed(mode * 7 + N)
        base = mode % 3
        if base == 0:
            A, B, aux = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev) * 0.05, torch.randn(N, device=dev)
            shape = (M, N)
        elif base == 1:
            A, B, aux = torch.randn(M, N, device=dev), torch.randn(K, N, device=dev) * 0.05, None
            shape = (M, K)
        else:
            A, B, aux = torch.randn(M, K, device=dev), torch.randn(M, N, device=dev), None
            shape = (K, N)
        H0 = torch.randn(*shape, device=dev)
        ref, bad = None, 0
        for rep in range(30):
            outs = []
            for c in range(2):
                with torch.cuda.stream(streams[c]):
                    C = H0.clone()
                    db = torch.zeros(N, device=dev) if base == 2 else aux
                    ctxs[c].dbg_gemm(mode, A, B, C, db, M, N, K, 1)
                    outs.append(C)
            torch.cuda.synchronize()
            for C in outs:
                if ref is None:
                    ref = C.clone()
                elif not torch.equal(C, ref):
                    bad += 1
                    if bad == 1:
                        d = (C - ref).abs()
                        print(f"   first diff: max {d.max().item():.3e} nnz {(d > 0).sum().item()} rows {(d > 0).any(1).sum().item()}")
        print(f"mode {mode} N={N} K={K}: {bad} differing of 59")
