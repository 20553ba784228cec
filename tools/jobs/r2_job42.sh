#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job42; mkdir -p "$OUT"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'checksum', d.get('checksum_ok'))
PY
}
for rep in 1 2; do
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12.json" "k12"
KPAL_QUAD_STEPS=8 python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_s8.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_s8.json" "k12 forced 8 (no sample)"
done
for k in 11 13 15; do python3 bench.py --k $k --steps 4 --warmup 1 --no-cpu > "$OUT/bench_k$k.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k$k.json" "k$k"; done
KPAL_QUAD_VERBOSE=1 python3 tools/skewbench.py 2>&1 | grep -v amdgpu.ids | grep -v "hot-table" | uniq | tail -30
( timeout 600 python -m pytest tests/test_gpu_count.py -m gpu -x -q -k "skew or tile or halve" > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" ); tail -3 "$OUT/pytest_count.log"
