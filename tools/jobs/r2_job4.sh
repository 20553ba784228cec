#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job4; mkdir -p "$OUT"
hipcc -O3 --offload-arch=gfx950 -o /tmp/store_probe2 tools/store_probe2.hip && timeout 600 /tmp/store_probe2 > "$OUT/store_probe2.log" 2>&1
cat "$OUT/store_probe2.log"
