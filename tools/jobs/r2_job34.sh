#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job34; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_count.py -m gpu -x -q > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" ); tail -3 "$OUT/pytest_count.log"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'checksum', d.get('checksum_ok'))
PY
}
for rep in 1 2; do
export KPAL_HIP_LIBRARY=$ROOT/build_ab/libkpal_hip_pair16.so
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_pair16.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_pair16.json" "k12 2048 rows x 16 (paired)"
unset KPAL_HIP_LIBRARY
KPAL_QUAD_VERBOSE=1 python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12.json" 2> "$OUT/bench_k12.err"; show "$OUT/bench_k12.json" "k12 1024 rows x 32"; grep "steps per wave\|steps/wave" "$OUT/bench_k12.err" | sort | uniq -c | head -5
done
KPAL_QUAD_STEPS=7 python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_s7.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_s7.json" "k12 1024 x 32 steps 7"
KPAL_QUAD_STEPS=6 python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_s6.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_s6.json" "k12 1024 x 32 steps 6"
for k in 11 9 13 15; do
  KPAL_QUAD_VERBOSE=1 python3 bench.py --k $k --steps 4 --warmup 1 --no-cpu > "$OUT/bench_k${k}.json" 2> "$OUT/bench_k$k.err"; show "$OUT/bench_k${k}.json" "k$k"; grep "steps per wave" "$OUT/bench_k$k.err" | sort | uniq -c | head -4
done
python3 tools/skewbench.py 2>&1 | grep -v amdgpu.ids | tail -5
python3 tools/skewbench.py --k 11 2>&1 | grep -v amdgpu.ids | tail -5
grep -v amdgpu.ids "$OUT/bench.err" | tail -5
