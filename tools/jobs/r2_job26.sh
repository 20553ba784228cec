#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job26; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for k in 12 11; do
rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_VALU -d "$OUT/pmc_lds_k$k" -o l -- python3 "$ROOT/bench.py" --no-cpu --k $k --reads 20000000 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_lds_k$k.err"
rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d "$OUT/pmc_wave_k$k" -o v -- python3 "$ROOT/bench.py" --no-cpu --k $k --reads 20000000 --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_wave_k$k.err"
python3 "$ROOT/tools/pmc_counters.py" "k=$k" "$OUT/pmc_lds_k$k" "$OUT/pmc_wave_k$k" > "$OUT/pmc_k$k.json"
done
find "$OUT" -name '*.db' -delete; find "$OUT" -name '*counter_collection.csv' -size +20M -delete
cat "$OUT/pmc_k12.json" | head -c 6000
