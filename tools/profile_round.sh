#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round.sh r1'): rocprofv3 kernel stats of the
# default bench command, and the FETCH_SIZE / WRITE_SIZE passes, into gpurun_out/<round>/.
set -u
ROUND=${1:-r1}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 10 --warmup 2 > "$OUT/bench_k12_n1.json" 2> "$OUT/bench_k12_n1.err"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_k12" -o k12 -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 > "$OUT/bench_k12_n1_under_rocprof.json" 2> "$OUT/stats_k12.err"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch_k12" -o f -- python3 "$ROOT/bench.py" --no-cpu --reads 20000000 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_fetch_k12.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write_k12" -o w -- python3 "$ROOT/bench.py" --no-cpu --reads 20000000 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_write_k12.err"
# 3 steps (1 warm-up + 2) of 20 M reads x 151 bytes
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch_k12" "$OUT/pmc_write_k12" 9060000000 \
  "rocprofv3 --output-format csv --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around python3 bench.py --no-cpu --reads 20000000 --k 12 --steps 2 --warmup 1; KB per dispatch averaged over dispatches; gfx950 correction: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)" chunk_scatter > "$OUT/pmc_hbm_traffic.json"
find "$OUT/stats_k12" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_k12_kernel_stats.csv" \;
# BASELINE config 4 (k = 15)
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_k15" -o k15 -- python3 "$ROOT/bench.py" --k 15 --steps 3 --warmup 1 --no-cpu > "$OUT/bench_k15_n1_under_rocprof.json" 2> "$OUT/stats_k15.err"
find "$OUT/stats_k15" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_k15_kernel_stats.csv" \;
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch_k15" -o f -- python3 "$ROOT/bench.py" --no-cpu --reads 20000000 --k 15 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_fetch_k15.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write_k15" -o w -- python3 "$ROOT/bench.py" --no-cpu --reads 20000000 --k 15 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_write_k15.err"
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch_k15" "$OUT/pmc_write_k15" 9060000000 \
  "rocprofv3 --output-format csv --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around python3 bench.py --no-cpu --reads 20000000 --k 15 --steps 2 --warmup 1 (one batch of 3.02 GB per step); gfx950 correction: FETCH_SIZE x2" coarse_scatter > "$OUT/pmc_hbm_traffic_k15.json"
# keep only the small summaries (the merge back is capped at 64 MiB)
find "$OUT" -name '*.db' -delete; find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*counter_collection.csv' -size +20M -delete
ls -la "$OUT"
