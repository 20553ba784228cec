#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job46; mkdir -p "$OUT"
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], 'balance_tiled', round(d['roofline']['kernels_ms_per_step']['balance_tiled'],3), 'ms')
PY
}
for k in 13 15; do
for lib in real bprobe1 bprobe2; do
  if [ $lib = real ]; then unset KPAL_HIP_LIBRARY; else export KPAL_HIP_LIBRARY=$ROOT/build_ab/libkpal_hip_$lib.so; fi
  python3 bench.py --k $k --steps 3 --warmup 1 --no-cpu > "$OUT/bench_k${k}_$lib.json" 2>> "$OUT/bench.err"; show "$OUT/bench_k${k}_$lib.json" "k$k $lib"
done; done
