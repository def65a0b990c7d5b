"""Victim: k_l1fwd_mfma (through a one-hidden-layer rlx_mlp_fwd_f32) -> H1 captured from the scratch arena; aggressor: dbg_gemm."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, ROOT)
import torch
from rlx_amd.hip import Ctx, mlp_desc
from rlx_amd.hip import lib as L
dev = torch.device("cuda:0")
O, A, M = 17, 6, 32768
pd = mlp_desc(O, [512], A, L.ACT_ELU if hasattr(L, "ACT_ELU") else 1, True, True)
npar = O * 512 + 3 * 512 + 512 * A + A + A
torch.manual_seed(0)
P = torch.randn(npar, device=dev) * 0.1
P[O * 512 + 512: O * 512 + 1024] = 1.0 + 0.1 * torch.randn(512, device=dev)
ctxs = (Ctx(0), Ctx(0))
streams = (torch.cuda.current_stream(), torch.cuda.Stream())
X = torch.randn(M, O, device=dev)
class _Buf:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
agg = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, K = (128, 256) if agg in (1, 4) else (256, 512)
base = agg % 3
if base == 0: Aa, Bb, aux, shape = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev) * .05, torch.randn(N, device=dev), (M, N)
elif base == 1: Aa, Bb, aux, shape = torch.randn(M, N, device=dev), torch.randn(K, N, device=dev) * .05, None, (M, K)
else: Aa, Bb, aux, shape = torch.randn(M, K, device=dev), torch.randn(M, N, device=dev), torch.zeros(N, device=dev), (K, N)
out = torch.empty(M, A, device=dev)
ctxs[0].mlp_fwd(pd, P, X, out); torch.cuda.synchronize()
H1 = torch.as_tensor(_Buf(ctxs[0].get_counter("scratch_ptr:0:23"), M * 512), device=dev).view(M, 512)
ref = H1.clone()
bad = 0
for rep in range(60):
    with torch.cuda.stream(streams[1]):
        for _ in range(3):
            C = torch.zeros(*shape, device=dev)
            ctxs[1].dbg_gemm(agg, Aa, Bb, C, aux, M, N, K, 1)
    with torch.cuda.stream(streams[0]):
        ctxs[0].mlp_fwd(pd, P, X, out)
    torch.cuda.synchronize()
    if not torch.equal(H1, ref):
        bad += 1
        if bad <= 2:
            d = (H1 != ref)
            rows = d.any(1).nonzero().flatten().tolist()
            r = rows[0]; c0 = int(d[r].nonzero().flatten()[0])
            # candidates: same columns of the row one persistent-loop step earlier; the pre-activation from a float64 recompute
            W1 = P[:O * 512].view(O, 512).double(); b1 = P[O * 512:O * 512 + 512].double()
            gm = P[O * 512 + 512:O * 512 + 1024].double(); be = P[O * 512 + 1024:O * 512 + 1536].double()
            def fwd(xrow):
                z = xrow.double() @ W1 + b1
                y = (z - z.mean()) / torch.sqrt(z.var(unbiased=False) + 1e-6) * gm + be
                return torch.where(y > 0, y, torch.expm1(y)), z
            h_ok, z_ok = fwd(X[r])
            print("  bad :", [f"{v:.5f}" for v in H1[r, c0:c0 + 6].tolist()])
            print("  ref :", [f"{v:.5f}" for v in ref[r, c0:c0 + 6].tolist()])
            print("  f64 :", [f"{v:.5f}" for v in h_ok[c0:c0 + 6].tolist()])
            for dr in (-512 * 32, 512 * 32, -1, 1, -4, 4, -8, 8, -16, 16):
                if 0 <= r + dr < M:
                    print(f"  ref row{dr:+d}:", [f"{v:.5f}" for v in ref[r + dr, c0:c0 + 6].tolist()])
            print(f"rep {rep}: {len(rows)} rows differ:", [(r, r // 32, r % 32, int(d[r].sum()), d[r].nonzero().flatten()[:3].tolist(),
                                                              f"{(H1[r] - ref[r]).abs().max().item():.2e}") for r in rows[:8]])
print(f"aggressor mode {agg}: {bad} differing H1 of 60")
