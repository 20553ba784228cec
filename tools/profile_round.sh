#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/profile_round.sh r4'): the round's measurement artefacts into
# gpurun_out/<round>/ -- copy what should be judged into profiles/<round>/.
#   pmc_hbm_traffic.json / _k15        FETCH_SIZE / WRITE_SIZE passes (separate), per kernel; taken FIRST and copied to
#                                      profiles/<round>/ so that the bench lines of the same call report roofline.traffic
#   pmc_lds_quad.json / _k15           SQ LDS / VALU / VMEM counters of the counting kernels
#   bench_k12_n1.json                  the default bench command: headline + extra (config 4, config 5, end to end) + cpu_baseline
#   bench_k12_kernel_stats.csv         rocprofv3 --kernel-trace --stats of the headline alone (--no-extra: the extras launch
#                                      kernels of the same names on other sizes and would blur the averages)
#   bench_k15_*                        BASELINE config 4 as its own command, + kernel stats
#   matrix_k12_P64_*                   BASELINE config 5 (multiset prod, euclidean) as its own command, + kernel stats
set -u
ROUND=${1:-r6}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p "$OUT" "$ROOT/profiles/$ROUND"
export KPAL_HEAD=$(cat "$ROOT/.head" 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
pmc() {   # name, counters..., then -- bench args
  local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --output-format csv --pmc "${ctrs[@]}" -d "$OUT/$name" -o p -- python3 "$B" --no-cpu --no-extra "$@" > /dev/null 2> "$OUT/$name.err"
}
# ---- k = 12 counters: 3 steps (1 warm-up + 2) of 20 M reads x 151 bytes, one quad_scatter + one quad_hist launch per step
K12="--reads 20000000 --steps 2 --warmup 1"
pmc pmc_fetch_k12 FETCH_SIZE -- $K12
pmc pmc_write_k12 WRITE_SIZE -- $K12
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch_k12" "$OUT/pmc_write_k12" 9060000000 \
  "rocprofv3 --output-format csv --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around python3 bench.py --no-cpu --no-extra $K12 (k=12); KB per dispatch averaged over dispatches; gfx950 correction: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)" quad_scatter > "$OUT/pmc_hbm_traffic.json"
cp "$OUT/pmc_hbm_traffic.json" "$ROOT/profiles/$ROUND/pmc_hbm_traffic.json"
pmc pmc_lds_k12 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_VALU -- $K12
pmc pmc_wave_k12 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -- $K12
python3 "$ROOT/tools/pmc_counters.py" "rocprofv3 --pmc (two passes: LDS/VALU/VMEM instruction counters; wave-cycle breakdown) -- python3 bench.py --no-cpu --no-extra $K12 (k=12; 3.02 GB and one quad_scatter + one quad_hist launch per step)" "$OUT/pmc_lds_k12" "$OUT/pmc_wave_k12" > "$OUT/pmc_lds_quad.json"
cp "$OUT/pmc_lds_quad.json" "$ROOT/profiles/$ROUND/pmc_lds_quad.json"
# ---- k = 15 counters (40 M reads = 6.04 GB per step: AUTO takes the two-level quad pipeline once the feed is larger than half the 8 GiB table)
K15="--reads 40000000 --k 15 --steps 2 --warmup 1"
pmc pmc_fetch_k15 FETCH_SIZE -- $K15
pmc pmc_write_k15 WRITE_SIZE -- $K15
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch_k15" "$OUT/pmc_write_k15" 18120000000 \
  "rocprofv3 --output-format csv --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around python3 bench.py --no-cpu --no-extra $K15 (one batch of 6.04 GB per step); gfx950 correction: FETCH_SIZE x2" quad_scatter > "$OUT/pmc_hbm_traffic_k15.json"
cp "$OUT/pmc_hbm_traffic_k15.json" "$ROOT/profiles/$ROUND/pmc_hbm_traffic_k15.json"
pmc pmc_lds_k15 SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_VALU -- $K15
pmc pmc_wave_k15 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -- $K15
python3 "$ROOT/tools/pmc_counters.py" "rocprofv3 --pmc (two passes) -- python3 bench.py --no-cpu --no-extra $K15 (k=15, 6.04 GB per step)" "$OUT/pmc_lds_k15" "$OUT/pmc_wave_k15" > "$OUT/pmc_lds_quad_k15.json"
cp "$OUT/pmc_lds_quad_k15.json" "$ROOT/profiles/$ROUND/pmc_lds_quad_k15.json"
# ---- the bench lines (they read the counter profiles copied above)
python3 "$B" --steps 20 --warmup 5 > "$OUT/bench_k12_n1.json" 2> "$OUT/bench_k12_n1.err"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_k12" -o k12 -- python3 "$B" --steps 10 --warmup 2 --no-cpu --no-extra > "$OUT/bench_k12_n1_under_rocprof.json" 2> "$OUT/stats_k12.err"
find "$OUT/stats_k12" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_k12_kernel_stats.csv" \;
python3 "$B" --k 15 --steps 5 --warmup 1 --no-cpu --no-extra > "$OUT/bench_k15_n1.json" 2> "$OUT/bench_k15_n1.err"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_k15" -o k15 -- python3 "$B" --k 15 --steps 3 --warmup 1 --no-cpu --no-extra > "$OUT/bench_k15_n1_under_rocprof.json" 2> "$OUT/stats_k15.err"
find "$OUT/stats_k15" -name '*kernel_stats.csv' -exec cp {} "$OUT/bench_k15_kernel_stats.csv" \;
for k in 9 11 13 14; do python3 "$B" --k $k --steps 5 --warmup 1 --no-cpu --no-extra > "$OUT/bench_k${k}_n1.json" 2> "$OUT/bench_k${k}_n1.err"; done
# ---- BASELINE config 5 (64 profiles, k = 12)
python3 "$B" --workload matrix --steps 5 --warmup 1 > "$OUT/matrix_k12_P64_prod_bench.json" 2> "$OUT/matrix_prod.err"
python3 "$B" --workload matrix --metric euclidean --steps 5 --warmup 1 > "$OUT/matrix_k12_P64_euclidean_bench.json" 2> "$OUT/matrix_eucl.err"
python3 "$B" --workload matrix --metric sum --steps 3 --warmup 1 > "$OUT/matrix_k12_P64_sum_bench.json" 2> "$OUT/matrix_sum.err"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_matrix" -o m -- python3 "$B" --workload matrix --steps 5 --warmup 1 --no-cpu > "$OUT/matrix_k12_P64_prod_under_rocprof.json" 2> "$OUT/stats_matrix.err"
find "$OUT/stats_matrix" -name '*kernel_stats.csv' -exec cp {} "$OUT/matrix_k12_P64_prod_kernel_stats.csv" \;
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_matrix_s" -o m -- python3 "$B" --workload matrix --metric sum --steps 5 --warmup 1 --no-cpu > "$OUT/matrix_k12_P64_sum_under_rocprof.json" 2> "$OUT/stats_matrix_s.err"
find "$OUT/stats_matrix_s" -name '*kernel_stats.csv' -exec cp {} "$OUT/matrix_k12_P64_sum_kernel_stats.csv" \;
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_matrix_e" -o m -- python3 "$B" --workload matrix --metric euclidean --steps 5 --warmup 1 --no-cpu > "$OUT/matrix_k12_P64_euclidean_under_rocprof.json" 2> "$OUT/stats_matrix_e.err"
find "$OUT/stats_matrix_e" -name '*kernel_stats.csv' -exec cp {} "$OUT/matrix_k12_P64_euclidean_kernel_stats.csv" \;
# ---- config 5 counters: where the staged loads of the matrix kernel come from (HBM bytes; L2 requests / hits / misses)
M5="--workload matrix --steps 2 --warmup 1 --no-cpu"
cd /tmp
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch_matrix" -o p -- python3 "$B" $M5 > /dev/null 2> "$OUT/pmc_fetch_matrix.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write_matrix" -o p -- python3 "$B" $M5 > /dev/null 2> "$OUT/pmc_write_matrix.err"
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch_matrix" "$OUT/pmc_write_matrix" 0 \
  "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around python3 bench.py $M5 (64 profiles, k = 12, multiset prod); gfx950 correction: FETCH_SIZE x2" matrix_rdiff_all > "$OUT/pmc_hbm_traffic_matrix.json"
rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_VALU -d "$OUT/pmc_sq_matrix" -o p -- python3 "$B" $M5 > /dev/null 2> "$OUT/pmc_sq_matrix.err"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/pmc_wave_matrix" -o p -- python3 "$B" $M5 > /dev/null 2> "$OUT/pmc_wave_matrix.err"
python3 "$ROOT/tools/pmc_counters.py" "rocprofv3 --pmc (two passes) -- python3 bench.py $M5 (64 profiles, k = 12, multiset prod: matrix_rdiff_all)" "$OUT/pmc_sq_matrix" "$OUT/pmc_wave_matrix" > "$OUT/pmc_sq_matrix.json"
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d "$OUT/pmc_l2_matrix" -o p -- python3 "$B" $M5 > /dev/null 2> "$OUT/pmc_l2_matrix.err"
python3 - "$OUT/pmc_l2_matrix" > "$OUT/pmc_l2_matrix.json" <<'PY'
import collections, csv, json, os, sys
per = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for d, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith('counter_collection.csv'):
            for row in csv.DictReader(open(os.path.join(d, f))):
                name = row['Kernel_Name'].split('(')[0].replace('void ', '')
                per[name][row['Counter_Name']] += float(row['Counter_Value']); disp[name].add(row['Dispatch_Id'])
json.dump({n: dict({k: v / len(disp[n]) for k, v in c.items()}, dispatches=len(disp[n])) for n, c in per.items() if 'matrix' in n}, sys.stdout, indent=1)
PY
# ---- vector kernels (north_star: "fused balance+multiset-distance reduction over paired profiles"): pair distance, the fused
#      balance + distance, balance, strand balance / split, summaries -- HIP-event times, rocprofv3 kernel stats, HBM counters
cd "$ROOT"
for k in 12 15; do python3 tools/vbench.py --k $k > "$OUT/vbench_k$k.log" 2>&1; done
cd /tmp
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_vec" -o v -- python3 "$ROOT/tools/vbench.py" --k 12 > /dev/null 2> "$OUT/stats_vec.err"
find "$OUT/stats_vec" -name '*kernel_stats.csv' -exec cp {} "$OUT/vbench_k12_kernel_stats.csv" \;
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch_vec" -o p -- python3 "$ROOT/tools/vbench.py" --k 12 > /dev/null 2> "$OUT/pmc_fetch_vec.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write_vec" -o p -- python3 "$ROOT/tools/vbench.py" --k 12 > /dev/null 2> "$OUT/pmc_write_vec.err"
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch_vec" "$OUT/pmc_write_vec" 0 \
  "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around python3 tools/vbench.py --k 12 (two Poisson(16.6) profiles of 4^12 bins resident in HBM); gfx950 correction: FETCH_SIZE x2" pair_distance_balanced > "$OUT/pmc_hbm_traffic_vec_k12.json"
# ---- the drop-in entry points on host-resident input (never the bench value): a multi-GB FASTA file through Profile.from_fasta /
#      the CLI / the shard cutter; lists of reads through Profile.from_sequences
cd "$ROOT"
python3 tools/clibench.py --gb 8 > "$OUT/clibench.json" 2> "$OUT/clibench.err"
python3 tools/hostbench.py --reads 20000000 2>&1 | head -8 > "$OUT/hostbench.log"
for t in 1 16; do KPAL_GATHER_THREADS=$t python3 tools/seqbench.py 2>&1 | grep -v amdgpu; done > "$OUT/seqbench.log"
python3 tools/clibench.py --by-record > "$OUT/clibench_by_record.json" 2> "$OUT/clibench_by_record.err"
# ---- skewed inputs
python3 tools/skewbench.py > "$OUT/skewbench_k12.log" 2>&1
python3 tools/skewbench.py --k 15 > "$OUT/skewbench_k15.log" 2>&1
python3 tools/skewdiag.py > "$OUT/skewdiag_k12.log" 2> /dev/null
python3 tools/skewdiag.py --k 15 > "$OUT/skewdiag_k15.log" 2> /dev/null
# ---- what a vector instruction costs to issue (round 6: the scatter's opcode mix priced with measured costs)
hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_bench tools/valu_bench.hip 2> /dev/null && timeout 300 /tmp/valu_bench > "$OUT/valu_bench.log" 2>&1
# ---- LDS primitives (what an LDS atomic costs with and without bank conflicts)
hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_bench tools/lds_bench.hip 2> /dev/null && timeout 300 /tmp/lds_bench > "$OUT/lds_bench.log" 2>&1
# ---- the N > 1 path of bench.py with 2, 3, 4 and 8 real ranks on this one GPU over the test stand-in for RCCL, synchronous and
#      asynchronous calls (correctness only: merged_equals_single_stream of every merge mode)
mkdir -p gpurun_out/multi_rank
{ bash tools/multi_rank_one_gpu.sh; echo "# KPAL_FAKE_RCCL_ASYNC=1 KPAL_FAKE_RCCL_DELAY_MS=5"; KPAL_FAKE_RCCL_ASYNC=1 KPAL_FAKE_RCCL_DELAY_MS=5 bash tools/multi_rank_one_gpu.sh; } > "$OUT/multi_rank_one_gpu.log" 2>&1
# keep only the small summaries (the merge back is capped at 64 MiB)
find "$OUT" -name '*.db' -delete; find "$OUT" -name '*kernel_trace.csv' -delete; find "$OUT" -name '*counter_collection.csv' -delete
ls -la "$OUT"; for f in "$OUT"/bench_k1[25]_n1.json "$OUT"/matrix*_bench.json; do echo "== $f"; head -c 1200 "$f"; echo; done
