#!/bin/bash
# Compact register / LDS / occupancy table of the kernels of one csrc unit (hipcc -Rpass-analysis=kernel-resource-usage).
#   tools/kernel_resources.sh gemm_bx [extra hipcc flags]
unit=$1; shift
cd "$(dirname "$0")/../rl-x_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -fno-vectorize "$@" \
  -Rpass-analysis=kernel-resource-usage -c $unit.hip -o /tmp/kr_$unit.o 2>&1 |
awk '/Function Name:/ {name=$(NF-1)} /VGPRs:/ && !/Spill/ && !/AGPR/ {v=$(NF-1)} /AGPRs:/ {a=$(NF-1)} /VGPRs Spill|VGPR Spill/ {sp=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)} /LDS Size/ {lds=$(NF-1); print name, "vgpr", v, "agpr", a, "spill", sp, "scratch", sc, "occ", oc, "lds", lds}' |
while read n rest; do echo "$(echo $n | c++filt | cut -c1-90) $rest"; done
