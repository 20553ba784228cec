#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job20; mkdir -p "$OUT"
timeout 600 python3 tools/diag/quad2_bisect.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/quad2_bisect.log"
cd /tmp && export TMPDIR=/tmp
for k in 15 13 14; do
python3 "$ROOT/bench.py" --k $k --steps 3 --warmup 1 --no-cpu > "$OUT/bench_k$k.json" 2> "$OUT/bench.err"; tail -2 "$OUT/bench.err" | grep -v amdgpu
python3 -c "
import json,sys
d=json.load(open('$OUT/bench_k$k.json'))
print('k$k auto', round(d['value'],1), round(d['ms_per_step'],3), d['checksum_ok'], {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
python3 "$ROOT/bench.py" --k 16 --reads 20000000 --steps 3 --warmup 1 --no-cpu > "$OUT/bench_k16.json" 2> "$OUT/bench.err"; python3 -c "
import json,sys
d=json.load(open('$OUT/bench_k16.json'))
print('k16 20M reads auto', round(d['value'],1), round(d['ms_per_step'],3), d['checksum_ok'], {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
cd "$ROOT"
( timeout 1200 python -m pytest tests/test_gpu_count.py -m gpu -x -q --durations=6 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log" ); tail -12 "$OUT/pytest.log"
