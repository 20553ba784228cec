#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job8; mkdir -p "$OUT"
timeout 300 python3 tools/diag/quad_bisect.py > "$OUT/bisect8.log" 2>&1; cat "$OUT/bisect8.log"
KPAL_QUAD_WAVES=16 timeout 300 python3 tools/diag/quad_bisect.py > "$OUT/bisect16.log" 2>&1; cat "$OUT/bisect16.log"
cd /tmp && export TMPDIR=/tmp
for w in 8 16 8 16; do
  KPAL_QUAD_WAVES=$w python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu > "$OUT/bench_w$w.json" 2> "$OUT/bench_w$w.err"
  python3 -c "
import json,sys
d=json.load(open('$OUT/bench_w$w.json'))
print('waves $w', d['value'], d['ms_per_step'], d['checksum_ok'], d['roofline']['kernels_ms_per_step'])"
done
KPAL_QUAD_WAVES=8 python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --k 11 > "$OUT/bench_k11.json" 2>&1; python3 -c "
import json,sys
d=json.load(open('$OUT/bench_k11.json'))
print('k11', d['value'], d['ms_per_step'], d['checksum_ok'], d['roofline']['kernels_ms_per_step'])"
