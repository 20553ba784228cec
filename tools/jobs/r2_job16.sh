#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job16; mkdir -p "$OUT"
KPAL_QUAD_VERBOSE=0 timeout 300 python3 tools/diag/lowcomp.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/lowcomp_auto.log"
for s in 0 8; do
  KPAL_QUAD_STEPS=$s timeout 200 python3 tools/diag/quad_bisect.py > "$OUT/bisect_s$s.log" 2>&1; echo "steps $s: ok lines $(grep -c 'differing bins 0,' $OUT/bisect_s$s.log)"; grep -v "differing bins 0," "$OUT/bisect_s$s.log" | grep differing | head -5
done
KPAL_QUAD_VERBOSE=1 timeout 200 python3 tools/skewbench.py --strategy partition_quads 2>&1 | grep -v amdgpu.ids | tee "$OUT/skew_auto.log" | grep -v "kpal quad"
grep "kpal quad" "$OUT/skew_auto.log" | sort | uniq -c | head -12
( timeout 900 python -m pytest tests/test_gpu_count.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log" ); tail -5 "$OUT/pytest.log"
cd /tmp && export TMPDIR=/tmp
for s in 0 0; do
  python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu > "$OUT/bench_s$s.json" 2> "$OUT/bench.err"
  python3 -c "
import json,sys
d=json.load(open('$OUT/bench_s$s.json'))
print('steps $s', round(d['value'],1), round(d['ms_per_step'],3), d['checksum_ok'], {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
