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


def label(name):
    """rocprofv3 kernel name -> the label bench.py's roofline rows use (KERNEL_OF in bench.py)."""
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("rlx::", "")
    if short.startswith("k_gemm_bx<0"):
        return "k_gemm_bx<0,...>"
    if short.startswith("k_gemm_bx<1"):
        return "k_gemm_bx<1,...>"
    if short.startswith("k_dx_l1bwd<"):     # <NT, NW, ACT, LN, BX[, TWIN]>: the label carries the engine (5th argument)
        args = [a.strip() for a in short[short.index("<") + 1:].rstrip(">").split(",")]
        return "k_dx_l1bwd<..,true>" if len(args) >= 5 and args[4] == "true" else "k_dx_l1bwd<..,false>"
    return re.sub(r"<.*", "", short)


def per_kernel(dbpath, counter):
    """{label: {grid: [calls, counter sum, duration sum]}}; grid = workgroups of the launch (one per problem shape)."""
    cur = sqlite3.connect(dbpath).cursor()
    try:
        rows = cur.execute("select p.name, k.grid_x / k.workgroup_x, k.grid_y, count(*), sum(p.counter_value), sum(p.duration) "
                           "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id where p.counter_name = ? "
                           "group by p.name, k.grid_x, k.grid_y", (counter,)).fetchall()
    except sqlite3.Error:
        rows = [(n, 0, 1, c, t, d) for n, c, t, d in cur.execute(
            "select name, count(*), sum(counter_value), sum(duration) from pmc_events where counter_name = ? group by name",
            (counter,)).fetchall()]
    out = {}
    for name, gx, gy, calls, total, dur in rows:
        c = out.setdefault(label(name), {}).setdefault(f"{gx}x{gy}", [0, 0.0, 0.0])
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
           "kernels": {}, "rows": {}}
    print("| kernel | grid (workgroups) | launches | read MB/launch (corrected) | write MB/launch | HBM MB/launch | avg us (PMC pass) |")
    print("|---|---|---|---|---|---|---|")
    tot = lambda d: [sum(v[i] for v in d.values()) for i in range(3)]
    for k in sorted(f, key=lambda k: -tot(f[k])[1]):
        if keep and not any(s in k for s in keep):
            continue
        fc, ft, fd = tot(f[k])
        wc, wt, _ = tot(w[k]) if k in w else (0, 0.0, 0.0)
        rd, wr = ft * 1024 * 2 / fc, (wt * 1024 / wc if wc else 0.0)
        res["kernels"][k] = {"launches": fc, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                             "hbm_bytes_per_launch": round(rd + wr), "fetch_size_kb_raw_per_launch": ft / fc,
                             "write_size_kb_raw_per_launch": (wt / wc) if wc else None}
        res["rows"][k] = []
        for g in sorted(f[k], key=lambda g: -f[k][g][1]):
            c, t_, d_ = f[k][g]
            wg = w.get(k, {}).get(g)
            rdg, wrg = t_ * 1024 * 2 / c, (wg[1] * 1024 / wg[0] if wg and wg[0] else 0.0)
            res["rows"][k].append({"grid": g, "launches": c, "read_bytes_per_launch": round(rdg),
                                   "write_bytes_per_launch": round(wrg), "hbm_bytes_per_launch": round(rdg + wrg),
                                   "avg_us": round(d_ / c / 1e3, 2)})
            print(f"| {k} | {g} | {c} | {rdg/1e6:.2f} | {wrg/1e6:.2f} | {(rdg+wrg)/1e6:.2f} | {d_/c/1e3:.1f} |")
    json.dump(res, open(outp, "w"), indent=1)


if __name__ == "__main__":
    main()
