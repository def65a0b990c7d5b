"""One context repeats a split-fp16 kernel while another context keeps a DIFFERENT kernel running on a second stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import torch
from rlx_amd.hip import Ctx
dev = torch.device("cuda:0")
ctxs = (Ctx(0), Ctx(0))
streams = (torch.cuda.current_stream(), torch.cuda.Stream())
M = 32768
def operands(mode, N, K, seed):
    torch.manual_seed(seed)
    base = mode % 3
    if base == 0:
        return torch.randn(M, K, device=dev), torch.randn(K, N, device=dev) * 0.05, torch.randn(N, device=dev), (M, N)
    if base == 1:
        return torch.randn(M, N, device=dev), torch.randn(K, N, device=dev) * 0.05, None, (M, K)
    return torch.randn(M, K, device=dev), torch.randn(M, N, device=dev), torch.zeros(N, device=dev), (K, N)
for mode, N, K in ((4, 128, 256), (5, 128, 256), (5, 256, 512), (3, 256, 512)):
    for other, oN, oK in ((0, 256, 512), (2, 256, 512), (1, 256, 512), (3, 128, 256), (5, 128, 256)):
        A, B, aux, shape = operands(mode, N, K, 1)
        oA, oB, oaux, oshape = operands(other, oN, oK, 2)
        H0, oH0 = torch.randn(*shape, device=dev), torch.randn(*oshape, device=dev)
        ref, bad = None, 0
        for rep in range(25):
            with torch.cuda.stream(streams[1]):
                for _ in range(3):
                    oC = oH0.clone()
                    ctxs[1].dbg_gemm(other, oA, oB, oC, oaux, M, oN, oK, 1)
            with torch.cuda.stream(streams[0]):
                outs = []
                for _ in range(3):
                    C = H0.clone()
                    ctxs[0].dbg_gemm(mode, A, B, C, aux, M, N, K, 1)
                    outs.append(C)
            torch.cuda.synchronize()
            for C in outs:
                if ref is None:
                    ref = C.clone()
                elif not torch.equal(C, ref):
                    bad += 1
                    if bad == 1:
                        d = (C - ref).abs()
                        nz = (d > 0)
                        print(f"   first diff: max {d.max().item():.3e} nnz {nz.sum().item()} rows {nz.any(1).sum().item()} cols {nz.any(0).sum().item()}")
        print(f"mode {mode} ({N},{K}) next to mode {other} ({oN},{oK}): {bad} differing of 74")
