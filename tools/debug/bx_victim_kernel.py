"""Victim: exact-fp32 forward kernels (rlx_mlp_fwd_f32 / single GEMM) repeated; aggressor: one split-fp16 kernel on a 2nd stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from rlx_amd.hip import Ctx
import test_gpu_dist as T
dev = torch.device("cuda:0")
ps, cs, pd, cd, P0, C0 = T._nets(dev, seed=2)
ctxs = (Ctx(0), Ctx(0))
streams = (torch.cuda.current_stream(), torch.cuda.Stream())
M = 32768
X = torch.randn(M, T.O, device=dev)
torch.manual_seed(1)
aggs = {3: (torch.randn(M, 512, device=dev), torch.randn(512, 256, device=dev) * 0.05, torch.randn(256, device=dev), (M, 256), 256, 512),
        4: (torch.randn(M, 128, device=dev), torch.randn(256, 128, device=dev) * 0.05, None, (M, 256), 128, 256),
        5: (torch.randn(M, 512, device=dev), torch.randn(M, 256, device=dev), torch.zeros(256, device=dev), (512, 256), 256, 512),
        2: (torch.randn(M, 512, device=dev), torch.randn(M, 256, device=dev), torch.zeros(256, device=dev), (512, 256), 256, 512)}
vA, vB, vb = torch.randn(M, 512, device=dev), torch.randn(512, 256, device=dev) * 0.05, torch.randn(256, device=dev)
for l1mfma in (1,):
    ctxs[0].set_option("l1fwd_mfma", l1mfma)
    for victim in ("mlp_fwd",):
        for agg in (5, 4, 3, 2):
            A, B, aux, shape, N, K = aggs[agg]
            ref, bad = None, 0
            for rep in range(30):
                with torch.cuda.stream(streams[1]):
                    for _ in range(4):
                        C = torch.zeros(*shape, device=dev)
                        ctxs[1].dbg_gemm(agg, A, B, C, aux, M, N, K, 1)
                with torch.cuda.stream(streams[0]):
                    outs = []
                    for _ in range(3):
                        if victim == "mlp_fwd":
                            out = torch.empty(M, T.A, device=dev)
                            ctxs[0].mlp_fwd(pd, P0, X, out)
                        else:
                            out = torch.empty(M, 256, device=dev)
                            ctxs[0].dbg_gemm(0, vA, vB, out, vb, M, 256, 512, 1)
                        outs.append(out)
                torch.cuda.synchronize()
                for o in outs:
                    if ref is None:
                        ref = o.clone()
                    elif not torch.equal(o, ref):
                        bad += 1
                        if bad == 1:
                            d = (o - ref).abs(); nz = d > 0
                            print(f"    first diff: max {d.max().item():.3e} nnz {int(nz.sum())} rows {int(nz.any(1).sum())}")
            print(f"l1fwd_mfma={l1mfma} victim {victim} next to dbg_gemm mode {agg}: {bad} differing of 89")
