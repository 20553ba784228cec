#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job33; mkdir -p "$OUT"
bash tools/profile_round.sh r2 > "$OUT/profile.log" 2>&1; tail -5 "$OUT/profile.log" | cut -c1-300
cd "$ROOT"
python3 tools/skewbench.py > "$ROOT/gpurun_out/r2/skewbench_k12.log" 2>&1
python3 tools/skewbench.py --k 13 > "$ROOT/gpurun_out/r2/skewbench_k13.log" 2>&1
python3 tools/diag/prefix.py > "$ROOT/gpurun_out/r2/shared_prefix_k12.log" 2>&1
for k in 9 11 13 14; do python3 bench.py --k $k --steps 4 --warmup 1 --no-cpu > "$ROOT/gpurun_out/r2/bench_k${k}_n1.json" 2>/dev/null; done
python3 bench.py --steps 20 --warmup 5 > "$ROOT/gpurun_out/r2/bench_k12_n1_after_profile.json" 2> /dev/null; cut -c1-600 "$ROOT/gpurun_out/r2/bench_k12_n1_after_profile.json"
