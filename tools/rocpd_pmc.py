#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (rocpd sqlite): FETCH_SIZE and WRITE_SIZE, collected in
SEPARATE passes (TCC slots: FETCH_SIZE costs 3 of 4, WRITE_SIZE 2) with --kernel-trace only.
Corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": both counters are in KILOBYTES (expression /1024);
on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, i.e. reports half the bytes of wide coalesced reads ->
read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE is taken as is (* 1024; uncalibrated per the guide).
Usage: python tools/rocpd_pmc.py <fetch.db> <write.db> <out.json> [kernel-name-substring ...]"""
import json
import re
import sqlite3
import sys


def per_kernel(dbpath, counter):
    cur = sqlite3.connect(dbpath).cursor()
    rows = cur.execute("select name, count(*), sum(counter_value), sum(duration) from pmc_events where counter_name = ? "
                       "group by name", (counter,)).fetchall()
    out = {}
    for name, calls, total, dur in rows:
        short = re.sub(r"\(.*", "", name).replace("void ", "").replace("rlx::", "")
        # the split-bf16 kernels report under the kernel KIND bench.py's roofline uses (k_gemm_bx<0,...> = forward,
        # k_gemm_bx<1,...> = input gradient, k_gemm_dw_bx = weight gradient)
        if short.startswith("k_gemm_bx<0"):
            short = "k_gemm_fwd"
        elif short.startswith("k_gemm_bx<1"):
            short = "k_gemm_dx"
        elif short.startswith("k_gemm_dw_bx"):
            short = "k_gemm_dw"
        short = re.sub(r"<.*", "", short)
        c = out.setdefault(short, [0, 0.0, 0.0])
        c[0] += calls
        c[1] += total
        c[2] += dur
    return out


def main():
    fetch, write, outp = sys.argv[1:4]
    keep = sys.argv[4:]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), bench.py --steps 1 --warmup 1",
           "corrections": "KB -> bytes (x1024); gfx950 FETCH_SIZE x2 (128-B requests tallied at 64 B); WRITE_SIZE as is",
           "kernels": {}}
    print("| kernel | launches | read MB/launch (corrected) | write MB/launch | HBM MB/launch | avg us (PMC pass) |")
    print("|---|---|---|---|---|---|")
    for k in sorted(f, key=lambda k: -f[k][1]):
        if keep and not any(s in k for s in keep):
            continue
        calls = f[k][0]
        rd = f[k][1] * 1024 * 2 / calls
        wr = (w[k][1] * 1024 / w[k][0]) if k in w and w[k][0] else 0.0
        res["kernels"][k] = {"launches": calls, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                             "hbm_bytes_per_launch": round(rd + wr), "fetch_size_kb_raw_per_launch": f[k][1] / calls,
                             "write_size_kb_raw_per_launch": (w[k][1] / w[k][0]) if k in w and w[k][0] else None}
        print(f"| {k} | {calls} | {rd/1e6:.2f} | {wr/1e6:.2f} | {(rd+wr)/1e6:.2f} | {f[k][2]/calls/1e3:.1f} |")
    json.dump(res, open(outp, "w"), indent=1)


if __name__ == "__main__":
    main()
