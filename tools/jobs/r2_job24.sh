#!/bin/bash
# A/B of the leaner quad scatter (scalar wave index, byte counters, zero-behind flush, shift-or encode) against the
# previous build (build_ab/libkpal_hip_base.so), after the count parity tests.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job24; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_count.py tests/test_gpu_fasta.py -m gpu -x -q > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" ); tail -5 "$OUT/pytest_count.log"
for lib in base new new; do
  if [ $lib = base ]; then export KPAL_HIP_LIBRARY=$ROOT/build_ab/libkpal_hip_base.so; else unset KPAL_HIP_LIBRARY; fi
  python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_$lib.json" 2>> "$OUT/bench.err"
  python3 - "$OUT/bench_k12_$lib.json" $lib <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'k12', round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', d['roofline']['kernels_ms_per_step'], 'checksum', d.get('checksum_ok'))
PY
done
for k in 11 13 15 16; do
  for lib in base new; do
    if [ $lib = base ]; then export KPAL_HIP_LIBRARY=$ROOT/build_ab/libkpal_hip_base.so; else unset KPAL_HIP_LIBRARY; fi
    python3 bench.py --k $k --steps 4 --warmup 1 --no-cpu > "$OUT/bench_k${k}_$lib.json" 2>> "$OUT/bench.err"
    python3 - "$OUT/bench_k${k}_$lib.json" $lib $k <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'k'+sys.argv[3], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', d['roofline']['kernels_ms_per_step'], 'checksum', d.get('checksum_ok'))
PY
  done
done
tail -5 "$OUT/bench.err"
python3 tools/skewbench.py > "$OUT/skewbench_k12.log" 2>&1; tail -12 "$OUT/skewbench_k12.log"
