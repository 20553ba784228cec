#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job45; mkdir -p "$OUT"
for seed in 21 22 23 24; do timeout 400 python3 tests/stress_count.py --seconds 140 --seed $seed > "$OUT/stress_$seed.log" 2>&1; echo "seed $seed rc=$?"; grep -v amdgpu.ids "$OUT/stress_$seed.log" | tail -2; done
