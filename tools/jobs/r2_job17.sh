#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job17; mkdir -p "$OUT"
KPAL_QUAD_VERBOSE=1 timeout 300 python3 tools/diag/prefix.py 2>&1 | grep -v amdgpu.ids | uniq | tee "$OUT/prefix_auto.log"
for s in 12 6; do echo "== steps $s"; KPAL_QUAD_STEPS=$s timeout 300 python3 tools/diag/prefix.py 2>&1 | grep -v amdgpu.ids | grep quads | tee "$OUT/prefix_s$s.log"; done
