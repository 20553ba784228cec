#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round.sh r2'): the round's measurement artefacts into
# gpurun_out/<round>/ -- copy what should be judged into profiles/<round>/.
#   bench_k12_n1.json                  default bench command (cpu_baseline included)
#   bench_k12_kernel_stats.csv         rocprofv3 --kernel-trace --stats of the same command
#   pmc_hbm_traffic.json               FETCH_SIZE / WRITE_SIZE passes (separate), per kernel
#   pmc_lds_quad.json                  SQ LDS / VALU / VMEM counters of the dominant kernels
#   bench_k15_* / pmc_hbm_traffic_k15  BASELINE config 4
#   matrix_k12_P64_*                   BASELINE config 5 (multiset prod, euclidean on the matrix cores)
set -u
ROUND=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT"
export KPAL_HEAD=$(cat "$ROOT/.head" 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp
# counter passes first: the bench lines below then report roofline.traffic from THIS build's profile
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch_k12" -o f -- python3 "$ROOT/bench.py" --no-cpu --reads 20000000 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_fetch_k12.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write_k12" -o w -- python3 "$ROOT/bench.py" --no-cpu --reads 20000000 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_write_k12.err"
# 3 steps (1 warm-up + 2) of 20 M reads x 151 bytes, one launch of quad_scatter per step
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch_k12" "$OUT/pmc_write_k12" 9060000000 \
  "rocprofv3 --output-format csv --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around python3 bench.py --no-cpu --reads 20000000 --k 12 --steps 2 --warmup 1; KB per dispatch averaged over dispatches; gfx950 correction: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)" quad_scatter > "$OUT/pmc_hbm_traffic.json"
rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_VALU -d "$OUT/pmc_lds" -o l -- python3 "$ROOT/bench.py" --no-cpu --reads 20000000 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_lds.err"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d "$OUT/pmc_wave" -o v -- python3 "$ROOT/bench.py" --no-cpu --reads 20000000 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_wave.err"
python3 "$ROOT/tools/pmc_counters.py" "rocprofv3 --pmc (two passes: LDS/VALU/VMEM instruction counters; wave-cycle breakdown) -- python3 bench.py --no-cpu --reads 20000000 --steps 2 --warmup 1 (k=12; 3.02 GB and one quad_scatter + one quad_hist launch per step)" "$OUT/pmc_lds" "$OUT/pmc_wave" > "$OUT/pmc_lds_quad.json"
mkdir -p "$ROOT/profiles/$ROUND" && cp "$OUT/pmc_hbm_traffic.json" "$ROOT/profiles/$ROUND/pmc_hbm_traffic.json"
python3 "$ROOT/bench.py" --steps 20 --warmup 5 > "$OUT/bench_k12_n1.json" 2> "$OUT/bench_k12_n1.err"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_k12" -o k12 -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu > "$OUT/bench_k12_n1_under_rocprof.json" 2> "$OUT/stats_k12.err"
find "$OUT/stats_k12" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_k12_kernel_stats.csv" \;
# BASELINE config 4 (k = 15)
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch_k15" -o f -- python3 "$ROOT/bench.py" --no-cpu --reads 40000000 --k 15 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_fetch_k15.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write_k15" -o w -- python3 "$ROOT/bench.py" --no-cpu --reads 40000000 --k 15 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_write_k15.err"
# (40 M reads = 6.04 GB per step: AUTO takes the two-level quad pipeline once the feed is larger than half the 8 GiB table)
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch_k15" "$OUT/pmc_write_k15" 18120000000 \
  "rocprofv3 --output-format csv --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around python3 bench.py --no-cpu --reads 40000000 --k 15 --steps 2 --warmup 1 (one batch of 6.04 GB per step); gfx950 correction: FETCH_SIZE x2" quad_scatter > "$OUT/pmc_hbm_traffic_k15.json"
cp "$OUT/pmc_hbm_traffic_k15.json" "$ROOT/profiles/$ROUND/pmc_hbm_traffic_k15.json"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_k15" -o k15 -- python3 "$ROOT/bench.py" --k 15 --steps 3 --warmup 1 --no-cpu > "$OUT/bench_k15_n1_under_rocprof.json" 2> "$OUT/stats_k15.err"
find "$OUT/stats_k15" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_k15_kernel_stats.csv" \;
# BASELINE config 5 (64 profiles, k = 12)
python3 "$ROOT/bench.py" --workload matrix --steps 5 --warmup 1 > "$OUT/matrix_k12_P64_prod_bench.json" 2> "$OUT/matrix_prod.err"
python3 "$ROOT/bench.py" --workload matrix --metric euclidean --steps 5 --warmup 1 > "$OUT/matrix_k12_P64_euclidean_bench.json" 2> "$OUT/matrix_eucl.err"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_matrix" -o m -- python3 "$ROOT/bench.py" --workload matrix --steps 5 --warmup 1 --no-cpu > "$OUT/matrix_k12_P64_prod_under_rocprof.json" 2> "$OUT/stats_matrix.err"
find "$OUT/stats_matrix" -name '*kernel_stats.csv' -exec cp {} "$OUT/matrix_k12_P64_prod_kernel_stats.csv" \;
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_matrix_e" -o m -- python3 "$ROOT/bench.py" --workload matrix --metric euclidean --steps 5 --warmup 1 --no-cpu > "$OUT/matrix_k12_P64_euclidean_under_rocprof.json" 2> "$OUT/stats_matrix_e.err"
find "$OUT/stats_matrix_e" -name '*kernel_stats.csv' -exec cp {} "$OUT/matrix_k12_P64_euclidean_kernel_stats.csv" \;
# keep only the small summaries (the merge back is capped at 64 MiB)
find "$OUT" -name '*.db' -delete; find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*counter_collection.csv' -size +20M -delete
ls -la "$OUT"; for f in "$OUT"/*.json; do echo "== $f"; head -c 1500 "$f"; echo; done
