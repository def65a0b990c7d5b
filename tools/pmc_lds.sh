#!/bin/bash
# LDS-side counters of the GEMM kernels (own PMC pass, --kernel-trace only).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=${1:-"python $REPO/tools/gemm_bench.py"}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_l
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace -d /tmp/pmc_l -- $CMD > /tmp/pmc_l.log 2>&1
python - <<PY
import sqlite3, glob, re
db = sqlite3.connect(glob.glob("/tmp/pmc_l/**/*.db", recursive=True)[0])
cur = db.cursor()
rows = cur.execute("select name, counter_name, count(*), sum(counter_value), sum(duration) from pmc_events group by name, counter_name").fetchall()
tab = {}
for name, c, n, v, d in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("rlx::", "")
    short = re.sub(r"<.*", "", short)
    t = tab.setdefault(short, {"n": n, "dur": d})
    t[c] = t.get(c, 0) + v
names = ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_BUSY_CU_CYCLES"]
print("| kernel | launches | avg us | " + " | ".join(names) + " | bank_conflict/idx_active |")
for k, t in sorted(tab.items(), key=lambda kv: -kv[1]["dur"])[:6]:
    n = t["n"]
    print(f"| {k} | {n} | {t['dur']/n/1e3:.1f} | " + " | ".join(f"{t.get(c,0)/n:.3g}" for c in names) + f" | {t.get('SQ_LDS_BANK_CONFLICT',0)/max(t.get('SQ_LDS_IDX_ACTIVE',1),1):.3f} |")
PY
