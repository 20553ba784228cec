#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job30; mkdir -p "$OUT"
( timeout 600 python -m pytest tests/test_gpu_count.py -m gpu -x -q -k "halve or skew" > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" ); tail -3 "$OUT/pytest_count.log"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', {k:round(v,2) for k,v in d['roofline']['kernels_ms_per_step'].items()}, 'checksum', d.get('checksum_ok'))
PY
}
for st in 7 6 7 6; do
KPAL_QUAD_STEPS=$st python3 bench.py --steps 10 --warmup 3 --no-cpu > "$OUT/bench_k12_s$st.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k12_s$st.json" "k12 steps $st"
done
for st in 7 6 4 3; do
echo "== forced steps $st"; KPAL_QUAD_STEPS=$st python3 tools/skewbench.py --strategy partition_quads 2>&1 | grep -v amdgpu.ids | tail -5
done
echo "== k11 forced"; for st in 7 6 4; do KPAL_QUAD_STEPS=$st python3 tools/skewbench.py --k 11 --strategy partition_quads 2>&1 | grep -v amdgpu.ids | tail -5; done
grep -v amdgpu.ids "$OUT/bench.err" | tail -5
