#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job43; mkdir -p "$OUT"
for seed in 11 12; do timeout 400 python3 tests/stress_count.py --seconds 150 --seed $seed > "$OUT/stress_$seed.log" 2>&1; echo "seed $seed rc=$?"; grep -v amdgpu.ids "$OUT/stress_$seed.log" | tail -3; done
timeout 300 python3 tests/stress_vec.py --seconds 60 > "$OUT/stress_vec.log" 2>&1; echo "vec rc=$?"; grep -v amdgpu.ids "$OUT/stress_vec.log" | tail -2
