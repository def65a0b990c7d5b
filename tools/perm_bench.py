#!/usr/bin/env python
"""Time of rlx_permutation_i32 (threefry bits + segmented stable radix sort, 3 rounds) at the PPO shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-x_amd"))
import torch
from rlx_amd.hip import Ctx
from rlx_amd.hip import lib as L
ctx = Ctx(0)
dev = torch.device("cuda:0")
for E, B in ((10, 524288), (10, 4194304), (10, 2048)):
    perm = torch.empty(E * B, dtype=torch.int32, device=dev)
    key = L.prng_key(1)
    for _ in range(2):
        ctx.permutation(key, perm, E, B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ctx.permutation(key, perm, E, B)
    torch.cuda.synchronize()
    print(f"E={E} B={B}: {1e3 * (time.perf_counter() - t0) / 5:.3f} ms per permutation")
