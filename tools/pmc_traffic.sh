#!/bin/bash
# HBM traffic of the bench kernels: two PMC passes (FETCH_SIZE, WRITE_SIZE), counters only with --kernel-trace.
# Run on the GPU box from the repo root: bash tools/pmc_traffic.sh   -> gpurun_out/pmc_traffic.{json,md}
set -x
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/pmc_w.log 2>&1
python $REPO/tools/rocpd_pmc.py $(find /tmp/pmc_f -name "*.db" | head -1) $(find /tmp/pmc_w -name "*.db" | head -1) \
    $REPO/gpurun_out/pmc_traffic.json > $REPO/gpurun_out/pmc_traffic.md
cat $REPO/gpurun_out/pmc_traffic.md
