#!/bin/bash
# MFMA-pipe utilisation of the MFMA kernels (SQ counters; own PMC pass, --kernel-trace only).
# Usage (GPU box, repo root): bash tools/pmc_mfma.sh ["command"]   -> gpurun_out/pmc_mfma.md
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=${1:-"python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_m
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_m -- $CMD > /tmp/pmc_m.log 2>&1
python - > $REPO/gpurun_out/pmc_mfma.md <<'PY'
import sqlite3, glob, re
db = sqlite3.connect(glob.glob("/tmp/pmc_m/**/*.db", recursive=True)[0])
cur = db.cursor()
rows = cur.execute("select name, counter_name, count(*), sum(counter_value), sum(duration) from pmc_events group by name, counter_name").fetchall()
tab = {}
for name, c, n, v, d in rows:
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("rlx::", "")
    short = re.sub(r"<.*", "", short)
    t = tab.setdefault(short, {})
    t[c] = t.get(c, 0) + v
    t["n_" + c] = t.get("n_" + c, 0) + n
    t["dur_" + c] = t.get("dur_" + c, 0) + d
print("MFMA-pipe counters per launch (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32")
print("SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace).  v_mfma_f32_32x32x2_f32 = 64 matrix-pipe cycles and 4096 FLOP per")
print("instruction and SIMD; v_mfma_f32_32x32x16_f16 (the *_bx kernels) = 32 cycles and 32768 fp16 FLOP, three of them per")
print("fp32-equivalent 32x32x16 product (= 10923 fp32-equivalent FLOP each).  The SQ counters come back for ONE of the 8 XCDs")
print("(calibration: tools/gemm_bench.py issues exactly M*N*K/2048 f32 MFMAs per launch, 8.0x the reported SQ_INSTS_MFMA), so the")
print("utilisation columns carry the factor 8: `busy` = 8 x SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs);")
print("`busy*` = 8 x SQ_INSTS_MFMA x (64 | 32) / (GRBM_GUI_ACTIVE x 1024) for pure f32 | fp16 kernels;")
print("last column: fp32-equivalent TFLOP/s from the instruction count (4096 | 10923 FLOP per MFMA).")
print()
print("| kernel | launches | avg us | GRBM_GUI_ACTIVE | SQ_INSTS_MFMA | MFMA_BUSY_CYCLES | MOPS_F32 | busy | busy* | TFLOP/s from INSTS_MFMA |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k, t in sorted(tab.items(), key=lambda kv: -kv[1].get("dur_GRBM_GUI_ACTIVE", 0)):
    n = t.get("n_GRBM_GUI_ACTIVE", 0)
    if not n or not t.get("SQ_INSTS_MFMA"):
        continue
    gui = t["GRBM_GUI_ACTIVE"] / n
    us = t["dur_GRBM_GUI_ACTIVE"] / n / 1e3
    insts = t["SQ_INSTS_MFMA"] / n
    busy = t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n
    bx = "_bx" in k or k.startswith("k_gemm_bx") or k.startswith("k_dx_l1bwd") or k.startswith("k_l1fwd_mfma")
    mixed = False
    cyc, fl = (32, 32768.0 / 3.0) if bx else (64, 4096.0)
    bstar = "n/a" if mixed else f"{8*insts*cyc/(gui*1024):.3f}"
    tf = "n/a" if mixed else f"{8*insts*fl/us/1e6:.1f}"
    print(f"| {k} | {n} | {us:.1f} | {gui:.4g} | {insts:.4g} | {busy:.4g} | {t.get('SQ_INSTS_VALU_MFMA_MOPS_F32',0)/n:.4g} | "
          f"{8*busy/(gui*1024):.3f} | {bstar} | {tf} |")
PY
cat $REPO/gpurun_out/pmc_mfma.md
grep -iE "error|fail|invalid" /tmp/pmc_m.log | head -5
