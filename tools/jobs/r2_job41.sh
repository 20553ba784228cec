#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job41; mkdir -p "$OUT"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'checksum', d.get('checksum_ok'))
PY
}
for rep in 1 2 3; do
export KPAL_HIP_LIBRARY=$ROOT/build_ab/libkpal_hip_item4b.so
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_item4.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_item4.json" "k12 4-byte items + bit-field hist"
unset KPAL_HIP_LIBRARY
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12.json" "k12 3-byte items"
done
export KPAL_HIP_LIBRARY=$ROOT/build_ab/libkpal_hip_item4b.so
KPAL_QUAD_STEPS=7 python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_item4_s7.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_item4_s7.json" "k12 4-byte items steps 7"
