#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table
(the equivalent of `--stats` CSV output): calls, total / avg / min / max duration, % of GPU time.
With --by-grid the rows are per (kernel, grid size): one row per problem shape of a kernel (forward layer 2 / layer 3 ...).
Usage: python tools/rocpd_stats.py <results.db> [--md] [--by-grid]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    by_grid = "--by-grid" in sys.argv
    grp = f"{name_col}, grid_x, grid_y" if by_grid else name_col
    sel = f"{name_col} || ' grid=' || (grid_x / workgroup_x) || 'x' || grid_y" if by_grid else name_col
    rows = cur.execute(f"select {sel}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {grp} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    span = cur.execute("select min(start), max(end) from kernels").fetchone()
    # union of the kernel intervals (kernels of different streams overlap)
    iv = cur.execute("select start, end from kernels order by start").fetchall()
    uni, cs, ce = 0, None, None
    for a, b in iv:
        if ce is None or a > ce:
            if ce is not None:
                uni += ce - cs
            cs, ce = a, b
        elif b > ce:
            ce = b
    if ce is not None:
        uni += ce - cs
    print(f"# kernels: {sum(r[1] for r in rows)} dispatches, sum of kernel durations {total/1e6:.2f} ms, "
          f"GPU busy (union of intervals) {uni/1e6:.2f} ms over a {(span[1]-span[0])/1e6:.2f} ms span")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for n, c, t, a, mn, mx in rows:
        gs = re.search(r" grid=\S+$", n)
        short = re.sub(r"\(.*", "", n) + (gs.group(0) if gs else "")
        short = short if len(short) < 110 else short[:107] + "..."
        print(f"| {short} | {c} | {t/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*t/total:.1f} |")


if __name__ == "__main__":
    main()
