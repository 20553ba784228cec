#!/bin/bash
# round 2, job 1: whole GPU suite (new config 4/5 tests), baseline bench at HEAD, LDS counters of chunk_scatter
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r2_job1
mkdir -p "$OUT"
export KPAL_HEAD=$(cat "$ROOT/.head" 2>/dev/null || echo unknown)
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log" )
tail -25 "$OUT/pytest.log"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 10 --warmup 2 > "$OUT/bench_k12.json" 2> "$OUT/bench_k12.err"
cat "$OUT/bench_k12.json"
rocprofv3 -L > "$OUT/counters.txt" 2>&1
grep -i "LDS" "$OUT/counters.txt" | head -40
rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -d "$OUT/pmc_lds" -o l -- python3 "$ROOT/bench.py" --no-cpu --reads 20000000 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_lds.err"
tail -3 "$OUT/pmc_lds.err"
python3 "$ROOT/tools/pmc_counters.py" "rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -- python3 bench.py --no-cpu --reads 20000000 --steps 2 --warmup 1 (k=12; 3.02 GB per step, 2 launches of chunk_scatter per step)" "$OUT/pmc_lds" > "$OUT/pmc_lds_chunk_scatter.json"
cat "$OUT/pmc_lds_chunk_scatter.json" | head -60
find "$OUT" -name '*.db' -delete; find "$OUT" -name '*counter_collection.csv' -size +20M -delete
