#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job21; mkdir -p "$OUT"
timeout 600 python3 tools/diag/quad2_bisect.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/quad2_bisect.log"
( timeout 1700 python -m pytest tests -m gpu -x -q --durations=10 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log" ); tail -18 "$OUT/pytest.log"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --k 16 --reads 20000000 --steps 3 --warmup 1 --no-cpu > "$OUT/bench_k16.json" 2> "$OUT/bench.err"; python3 -c "
import json,sys
d=json.load(open('$OUT/bench_k16.json'))
print('k16 20M reads auto', round(d['value'],1), round(d['ms_per_step'],3), d['checksum_ok'], {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
for k in 13 15; do echo "== skewbench k=$k"; timeout 300 python3 "$ROOT/tools/skewbench.py" --k $k 2>&1 | grep -v amdgpu.ids | tee "$OUT/skew_k$k.log"; done
