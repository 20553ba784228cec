#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job32; mkdir -p "$OUT"
KPAL_QUAD_VERBOSE=1 python3 tools/skewbench.py > "$OUT/skewbench_k12.log" 2>&1; grep -v amdgpu.ids "$OUT/skewbench_k12.log" | grep -v "hot-table\|expected backlog\|sample:" | tail -6
echo "== prefix"; KPAL_QUAD_VERBOSE=1 timeout 300 python3 tools/diag/prefix.py > "$OUT/prefix.log" 2>&1; grep -v amdgpu.ids "$OUT/prefix.log" | grep -v "hot-table\|expected backlog" | uniq | head -60
( timeout 600 python -m pytest tests/test_gpu_count.py -m gpu -x -q -k "skew or shared or prefix or low" > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" ); tail -3 "$OUT/pytest_count.log"
