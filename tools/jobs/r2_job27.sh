#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job27; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_count.py -m gpu -x -q -k "k15 or two_level or k16 or skew or strategies" > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" ); tail -3 "$OUT/pytest_count.log"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'checksum', d.get('checksum_ok'))
PY
}
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_w8.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_w8.json" "k12 8 waves"
KPAL_QUAD_WAVES=16 python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_w16.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_w16.json" "k12 16 waves x 6"
KPAL_QUAD_WAVES=16 KPAL_QUAD_STEPS=14 python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_w16s7.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_w16s7.json" "k12 16 waves x 7"
KPAL_QUAD_WAVES=16 python3 bench.py --k 11 --steps 5 --warmup 2 --no-cpu > "$OUT/bench_k11_w16.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k11_w16.json" "k11 16 waves x 6"
python3 bench.py --k 11 --steps 5 --warmup 2 --no-cpu > "$OUT/bench_k11_w8.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k11_w8.json" "k11 8 waves"
for k in 13 15; do
  python3 bench.py --k $k --steps 4 --warmup 1 --no-cpu > "$OUT/bench_k${k}_new.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k${k}_new.json" "new k$k"
done
python3 bench.py --k 16 --reads 240000000 --steps 2 --warmup 1 --no-cpu > "$OUT/bench_k16_new.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k16_new.json" "new k16 240M reads"
grep -v amdgpu.ids "$OUT/bench.err" | tail -5
