#!/usr/bin/env python
"""Print a window of a rocprofv3 kernel trace (rocpd sqlite) as a per-kernel timeline: start (us, relative), duration, queue,
kernel name -- enough to see which kernels of the two update streams overlap and where the launch gaps are.
Usage: python tools/rocpd_timeline.py <results.db> <start fraction of the trace, 0..1> <number of kernels>"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
name_col = "name" if "name" in cols else "kernel_name"
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = cur.execute(f"select {name_col}, start, end, {q or '0'} from kernels order by start").fetchall()
# find steady-state window: after 70% of the trace
t0 = rows[0][1]; t1 = rows[-1][2]
w0 = t0 + int(float(sys.argv[2]) * (t1 - t0))
out = [r for r in rows if r[1] >= w0][:int(sys.argv[3])]
base = out[0][1]
for n, s, e, qq in out:
    short = re.sub(r"\(.*", "", n).replace("void ", "").replace("rlx::", "")
    short = re.sub(r"<.*", "", short)[:28]
    print("%9.1f %8.1f  q%-3s %s" % ((s - base) / 1e3, (e - s) / 1e3, qq, short))
