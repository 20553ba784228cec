#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job29; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_count.py -m gpu -x -q > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" ); tail -3 "$OUT/pytest_count.log"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'checksum', d.get('checksum_ok'))
PY
}
for rep in 1 2; do
export KPAL_HIP_LIBRARY=$ROOT/build_ab/libkpal_hip_nopair.so
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_nopair.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_nopair.json" "k12 64-byte records alone"
unset KPAL_HIP_LIBRARY
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12.json" "k12 paired rows"
done
for k in 13 15; do
  python3 bench.py --k $k --steps 4 --warmup 1 --no-cpu > "$OUT/bench_k${k}.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k${k}.json" "k$k"
done
KPAL_QUAD_VERBOSE=1 python3 tools/skewbench.py > "$OUT/skewbench_k12.log" 2>&1; grep -v amdgpu.ids "$OUT/skewbench_k12.log" | grep -v "^\[kpal" | tail -6; grep "^\[kpal" "$OUT/skewbench_k12.log" | sort | uniq -c | head
python3 tools/skewbench.py --k 13 > "$OUT/skewbench_k13.log" 2>&1; grep -v amdgpu.ids "$OUT/skewbench_k13.log" | tail -6
python3 tools/skewbench.py --k 11 > "$OUT/skewbench_k11.log" 2>&1; grep -v amdgpu.ids "$OUT/skewbench_k11.log" | tail -6
grep -v amdgpu.ids "$OUT/bench.err" | tail -5
