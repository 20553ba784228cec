#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job19; mkdir -p "$OUT"
timeout 600 python3 tools/diag/quad2_bisect.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/quad2_bisect.log"
cd /tmp && export TMPDIR=/tmp
for strat in partition2_quads partition2; do
python3 "$ROOT/bench.py" --k 15 --steps 3 --warmup 1 --no-cpu --strategy $strat > "$OUT/bench_k15_$strat.json" 2> "$OUT/bench.err"; tail -2 "$OUT/bench.err"
python3 -c "
import json,sys
d=json.load(open('$OUT/bench_k15_$strat.json'))
print('k15 $strat', round(d['value'],1), round(d['ms_per_step'],3), d['checksum_ok'], {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
