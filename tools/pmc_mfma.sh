#!/bin/bash
# MFMA-pipe utilisation of the GEMM kernels (SQ counters; own pass, --kernel-trace only).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=${1:-"python $REPO/tools/gemm_bench.py"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_m
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_m -- $CMD > /tmp/pmc_m.log 2>&1
python - <<PY
import sqlite3, glob, re
db = sqlite3.connect(glob.glob("/tmp/pmc_m/**/*.db", recursive=True)[0])
cur = db.cursor()
rows = cur.execute("select name, counter_name, count(*), sum(counter_value), sum(duration) from pmc_events group by name, counter_name").fetchall()
tab = {}
for name, c, n, v, d in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("rlx::", "")
    short = re.sub(r"<.*", "", short)
    t = tab.setdefault(short, {"n": n, "dur": d})
    t[c] = t.get(c, 0) + v
print("| kernel | launches | avg us | GUI_ACTIVE cyc/launch | eff. clock GHz | MFMA busy / (GUI_ACTIVE*1024 SIMD) | WAIT_ANY/WAVE | WAIT_INST_ANY/WAVE | ACTIVE_INST/WAVE |")
print("|---|---|---|---|---|---|---|---|---|")
for k, t in sorted(tab.items(), key=lambda kv: -kv[1]["dur"]):
    if "GRBM_GUI_ACTIVE" not in t or not t["GRBM_GUI_ACTIVE"]:
        continue
    n = t["n"]; gui = t["GRBM_GUI_ACTIVE"] / n; us = t["dur"] / n / 1e3
    wave = max(t.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"| {k} | {n} | {us:.1f} | {gui:.0f} | {gui/us/1e3:.2f} | {t.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/n/(gui*1024):.3f} | "
          f"{t.get('SQ_WAIT_ANY',0)/wave:.2f} | {t.get('SQ_WAIT_INST_ANY',0)/wave:.2f} | {t.get('SQ_ACTIVE_INST_ANY',0)/wave:.2f} |")
PY
tail -8 /tmp/pmc_m.log | grep -v simple_timer
