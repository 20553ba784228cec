#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job44; mkdir -p "$OUT"
( timeout 1200 python -m pytest tests/test_gpu_count.py -m gpu -x -q -k "not full_size_k15" > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" ); tail -3 "$OUT/pytest_count.log"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'checksum', d.get('checksum_ok'))
PY
}
for rep in 1 2 3; do
export KPAL_HIP_LIBRARY=$ROOT/build_ab/libkpal_hip_bytes3.so
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_bytes.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_bytes.json" "k12 3 byte stores per item"
unset KPAL_HIP_LIBRARY
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12.json" "k12 dword OR + riders"
done
python3 tools/skewbench.py 2>&1 | grep -v amdgpu.ids | tail -5
