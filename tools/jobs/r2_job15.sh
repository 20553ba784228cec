#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job15; mkdir -p "$OUT"
KPAL_QUAD_VERBOSE=0 timeout 300 python3 tools/diag/lowcomp.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/lowcomp_auto.log"
echo "== steps 12 forced"; KPAL_QUAD_STEPS=12 timeout 300 python3 tools/diag/lowcomp.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/lowcomp_s12.log"
