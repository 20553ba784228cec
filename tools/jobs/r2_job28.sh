#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job28; mkdir -p "$OUT"
( timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log" ); tail -4 "$OUT/pytest_gpu.log"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'checksum', d.get('checksum_ok'))
PY
}
python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12.json" "k12"
KPAL_QUAD_STEPS=6 python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_s6.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_s6.json" "k12 steps 6"
for k in 9 11 13 14 15; do
  python3 bench.py --k $k --steps 4 --warmup 1 --no-cpu > "$OUT/bench_k${k}.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k${k}.json" "k$k"
done
KPAL_QUAD_STEPS=6 python3 bench.py --k 15 --steps 4 --warmup 1 --no-cpu > "$OUT/bench_k15_s6.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k15_s6.json" "k15 steps 6"
python3 tools/skewbench.py > "$OUT/skewbench_k12.log" 2>&1; grep -v amdgpu.ids "$OUT/skewbench_k12.log" | tail -6
python3 tools/skewbench.py --k 13 > "$OUT/skewbench_k13.log" 2>&1; grep -v amdgpu.ids "$OUT/skewbench_k13.log" | tail -6
grep -v amdgpu.ids "$OUT/bench.err" | tail -5
