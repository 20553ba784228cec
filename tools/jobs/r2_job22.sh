#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job22; mkdir -p "$OUT"
( timeout 300 python -m pytest tests/test_gpu_count.py -m gpu -x -q -k "k15_against_oracle" > "$OUT/pytest_k15.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_k15.log" ); tail -3 "$OUT/pytest_k15.log"
bash tools/profile_round.sh r2 > "$OUT/profile.log" 2>&1; tail -5 "$OUT/profile.log" | cut -c1-300
