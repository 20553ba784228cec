#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job13; mkdir -p "$OUT"
KPAL_QUAD_VERBOSE=1 timeout 200 python3 tools/skewbench.py --strategy partition_quads 2>&1 | grep -v amdgpu.ids | tee "$OUT/skew_auto.log"
for s in 12 8 6; do echo "== forced steps $s"; KPAL_QUAD_STEPS=$s KPAL_QUAD_VERBOSE=1 timeout 200 python3 tools/skewbench.py --strategy partition_quads 2>&1 | grep -v amdgpu.ids | grep -A4 "low-complexity\|AT-rich" | grep -v "^--" | tee -a "$OUT/skew_forced.log"; done
