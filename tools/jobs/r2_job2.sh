#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job2; mkdir -p "$OUT"
timeout 600 python3 tools/diag/k15_bisect.py 20000000 28000000 29000000 40000000 100000000 > "$OUT/k15_bisect.log" 2>&1
cat "$OUT/k15_bisect.log"
for b in store_bench store_bench2; do
  hipcc -O3 --offload-arch=gfx950 -o /tmp/$b tools/$b.hip && timeout 300 /tmp/$b > "$OUT/$b.log" 2>&1
  cat "$OUT/$b.log"
done
