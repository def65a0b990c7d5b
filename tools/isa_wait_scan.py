#!/usr/bin/env python
"""Which kernels wait with `s_waitcnt vmcnt(0)` INSIDE a loop that feeds the matrix pipe?  (CPU only: hipcc cross-compiles.)

A K loop that prefetches its operands is supposed to wait with a COUNTED vmcnt (leave the younger loads in flight).  hipcc falls back
to vmcnt(0) -- draining everything just requested -- when (1) a refill sits behind a condition, so the number of loads outstanding at the
loop header depends on the path, (2) the pointer is generic (loaded from a descriptor in memory, or laundered through `asm volatile`):
flat_load may return out of order with LDS traffic, (3) two load paths of different length meet inside the loop.  Round 6 found these
in seven kernels; this script lists, per kernel, the loop blocks that hold MFMAs and a vmcnt(0), plus flat loads.

    python tools/isa_wait_scan.py [unit ...]        # default: every rl-x_amd/csrc/*.hip
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rl-x_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-fno-slp-vectorize", "-fno-vectorize", "-S",
         "--cuda-device-only", "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]


def scan(path):
    name, cur, blocks, flat = None, None, {}, {}
    for ln in open(path):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, cur = m.group(1), None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", ln)
        if m:
            cur = (name, m.group(1))
            blocks[cur] = {"mfma": 0, "w0": 0, "loop": "Loop" in m.group(2)}
            continue
        if name and "flat_load" in ln:
            flat[name] = flat.get(name, 0) + 1
        if cur is None:
            continue
        b = blocks[cur]
        b["mfma"] += "v_mfma" in ln
        b["w0"] += bool(re.search(r"s_waitcnt vmcnt\(0\)", ln))
    agg = {}
    for (k, _), b in blocks.items():
        if b["loop"] and b["mfma"] and b["w0"]:
            a = agg.setdefault(k, [0, 0, 0])
            a[0] += 1; a[1] += b["mfma"]; a[2] += b["w0"]
    return agg, flat


units = sys.argv[1:] or sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(CSRC, "*.hip")))
for u in units:
    out = f"/tmp/isa_{u}.s"
    r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + [os.path.join(CSRC, u + ".hip"), "-o", out],
                       capture_output=True, text=True)
    if r.returncode:
        print(f"{u}: compile failed\n{r.stderr[-400:]}")
        continue
    agg, flat = scan(out)
    demangle = lambda k: subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:120]
    for k, a in sorted(agg.items(), key=lambda x: -x[1][2]):
        print(f"{u:10s} loop blocks {a[0]:3d}  mfma {a[1]:4d}  vmcnt(0) {a[2]:3d}  flat_load {flat.get(k, 0):3d}  {demangle(k)}")
    for k, n in flat.items():
        if k not in agg:
            print(f"{u:10s} flat_load {n:3d} (no MFMA loop with vmcnt(0))  {demangle(k)}")
