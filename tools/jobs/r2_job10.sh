#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job10; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "8 12" "8 14" "8 15" "8 16" "16 6" "8 12" "8 14" "8 15"; do
  set -- $cfg
  KPAL_QUAD_WAVES=$1 KPAL_QUAD_STEPS=$2 python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu > "$OUT/bench_w$1_s$2.json" 2> "$OUT/bench.err"
  python3 -c "
import json,sys
d=json.load(open('$OUT/bench_w$1_s$2.json'))
print('waves $1 steps $2', round(d['value'],1), round(d['ms_per_step'],3), d['checksum_ok'], {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
cd "$ROOT"
for s in 14 15 16; do
  KPAL_QUAD_STEPS=$s timeout 300 python3 tools/diag/quad_bisect.py > "$OUT/bisect_s$s.log" 2>&1; grep -c "differing bins 0," "$OUT/bisect_s$s.log"; grep -v "differing bins 0," "$OUT/bisect_s$s.log" | head -5
done
( KPAL_QUAD_STEPS=15 timeout 900 python -m pytest tests/test_gpu_count.py -m gpu -x -q -k "skew or overflow or g2 or g3 or every_k or mixed or unaligned or host_feed" > "$OUT/pytest_s15.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_s15.log" ); tail -5 "$OUT/pytest_s15.log"
