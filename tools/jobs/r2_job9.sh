#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job9; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_vec.py tests/test_gpu_options.py -m gpu -x -q --durations=5 > "$OUT/pytest_vec.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_vec.log" ); tail -12 "$OUT/pytest_vec.log"
bash tools/profile_round.sh r2a > "$OUT/profile.log" 2>&1; tail -60 "$OUT/profile.log" | cut -c1-600
