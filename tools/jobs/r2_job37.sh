#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job37; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_count.py -m gpu -x -q -k "two_level or k15 or k16 or skew or halve or strategies" > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" ); tail -3 "$OUT/pytest_count.log"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'checksum', d.get('checksum_ok'))
PY
}
for k in 13 15; do
  python3 bench.py --k $k --steps 4 --warmup 1 --no-cpu > "$OUT/bench_k${k}.json" 2> "$OUT/bench_k$k.err"; show "$OUT/bench_k${k}.json" "k$k"
done
for k in 13 14; do echo "== skew k=$k"; python3 tools/skewbench.py --k $k 2>&1 | grep -v amdgpu.ids | tail -5; done
echo "== k13 forced steps2"; for s2 in 8 6 4; do KPAL_QUAD_STEPS2=$s2 python3 tools/skewbench.py --k 13 2>&1 | grep -v amdgpu.ids | tail -5 | head -4; done
timeout 600 python3 tools/diag/quad2_bisect.py > "$OUT/quad2_bisect.log" 2>&1; tail -6 "$OUT/quad2_bisect.log"
