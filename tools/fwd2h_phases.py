#!/usr/bin/env python
"""Phase clock stamps of k_fwd2h (fwd2h.hip) for one workgroup: policy (376 -> 256 -> 256 -> 34) and critic (393 -> ... -> 1) shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from rlx_amd.hip import Ctx, mlp_desc
from oracle import sac
dev = torch.device("cuda:0")
ctx = Ctx(0)
rng = np.random.default_rng(0)
st = torch.zeros(16, dtype=torch.int64, device=dev)
for O, od, pol in ((376, 34, True), (393, 1, False)):
    ps, qs = sac.make_specs(O if pol else O - 17, 17, 256)
    spec = ps if pol else qs
    par = torch.from_numpy(sac.lecun_normal_init(spec, rng).astype(np.float32)).to(dev)
    d = mlp_desc(spec.in_dim, spec.hidden, spec.out_dim, spec.act, spec.ln_first, False)
    x = torch.randn(4096, O, device=dev)
    out = torch.empty(4096, od, device=dev)
    for _ in range(3): ctx.mlp_fwd(d, par, x, out)
    ctx.dbg_set_stamps(st)
    ctx.mlp_fwd(d, par, x, out); torch.cuda.synchronize()
    ctx.dbg_set_stamps(None)
    s = st.cpu().numpy().astype(np.int64)
    names = ["prologue+Wh", "X load+stage", "layer 1", "h1 epilogue", "layer 2", "h2 epilogue", "head"]
    print(f"O={O} OD={od}: " + ", ".join(f"{n} {int(s[i+1]-s[i])}" for i, n in enumerate(names)) + f"  total {int(s[7]-s[0])} ticks (clock64, 100 MHz => x10 ns)")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): ctx.mlp_fwd(d, par, x, out)
    torch.cuda.synchronize(); print(f"   {1e6*(time.perf_counter()-t0)/200:.1f} us per rlx_mlp_fwd_f32 call (wfrag + k_fwd2h)")

