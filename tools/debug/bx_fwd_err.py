"""Forward error of the split-operand engine against the exact-fp32 engine and float64 (numpy) on SAC-shaped nets at M = 4096."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from rlx_amd.hip import Ctx, mlp_desc, ACT_RELU
dev = torch.device("cuda:0")
ctx = Ctx(0)
rng = np.random.default_rng(0)
M = 4096
for (I, H, OUT, xs) in ((393, 256, 1, 1.0), (393, 256, 1, 3.0), (376, 256, 34, 1.0), (392, 256, 1, 1.0), (376, 256, 34, 10.0), (376, 256, 34, 0.05)):
    d = mlp_desc(I, [H, H], OUT, ACT_RELU, False, False)
    W0, b0 = rng.standard_normal((I, H)) / np.sqrt(I), 0.1 * rng.standard_normal(H)
    W1, b1 = rng.standard_normal((H, H)) / np.sqrt(H), 0.1 * rng.standard_normal(H)
    Wh, bh = 0.1 * rng.standard_normal((H, OUT)) / np.sqrt(H), 0.1 * rng.standard_normal(OUT)
    flat = np.concatenate([a.reshape(-1) for a in (W0, b0, W1, b1, Wh, bh)]).astype(np.float32)
    ld = (I + 3) & ~3
    x = np.zeros((M, ld), np.float32)
    x[:, :I] = xs * rng.standard_normal((M, I))
    f = lambda a: a.astype(np.float64)
    fl = f(flat)
    o = 0
    def take(n, shape):
        global o
        v = fl[o:o + n].reshape(shape); o += n
        return v
    w0, c0, w1, c1, wh, ch = take(I * H, (I, H)), take(H, (H,)), take(H * H, (H, H)), take(H, (H,)), take(H * OUT, (H, OUT)), take(OUT, (OUT,))
    h = np.maximum(f(x[:, :I]) @ w0 + c0, 0); h = np.maximum(h @ w1 + c1, 0); ref = h @ wh + ch
    P, X = torch.from_numpy(flat).to(dev), torch.from_numpy(x).to(dev)
    res = {}
    for bx in (1, 0):
        ctx.set_option("gemm_bx", bx)
        out = torch.empty(M, OUT, device=dev)
        ctx.mlp_fwd(d, P, X[:, :I].contiguous(), out)
        res[bx] = out.cpu().numpy().astype(np.float64)
    ctx.set_option("gemm_bx", 1)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    print(f"in {I} x{xs}: split engine vs f64 {rel(res[1], ref):.2e}, exact engine vs f64 {rel(res[0], ref):.2e}, max abs split {np.abs(res[1]-ref).max():.2e} exact {np.abs(res[0]-ref).max():.2e}")
