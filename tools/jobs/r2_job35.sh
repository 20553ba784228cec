#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job35; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_count.py -m gpu -x -q > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" ); tail -3 "$OUT/pytest_count.log"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'checksum', d.get('checksum_ok'))
PY
}
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12.json" 2> "$OUT/bench_k12.err"; show "$OUT/bench_k12.json" "k12"
for k in 13 14 15; do
  python3 bench.py --k $k --steps 4 --warmup 1 --no-cpu > "$OUT/bench_k${k}.json" 2> "$OUT/bench_k$k.err"; show "$OUT/bench_k${k}.json" "k$k"
done
python3 bench.py --k 16 --reads 240000000 --steps 2 --warmup 1 --no-cpu > "$OUT/bench_k16.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k16.json" "k16 240M reads"
timeout 600 python3 tools/diag/quad2_bisect.py > "$OUT/quad2_bisect.log" 2>&1; tail -12 "$OUT/quad2_bisect.log"
