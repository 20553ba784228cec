#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job11; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in "8 12 12" "8 12 6" "8 13 13" "8 14 7" "8 12 12" "8 13 13"; do
  set -- $cfg
  KPAL_QUAD_WAVES=$1 KPAL_QUAD_STEPS=$2 KPAL_QUAD_DEPTH=$3 python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu > "$OUT/bench_w$1_s$2_d$3.json" 2> "$OUT/bench.err"
  python3 -c "
import json,sys
d=json.load(open('$OUT/bench_w$1_s$2_d$3.json'))
print('waves $1 steps $2 depth $3', round(d['value'],1), round(d['ms_per_step'],3), d['checksum_ok'], {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
cd "$ROOT"
for s in 12 13 14; do
  KPAL_QUAD_STEPS=$s timeout 300 python3 tools/diag/quad_bisect.py > "$OUT/bisect_s$s.log" 2>&1; grep -c "differing bins 0," "$OUT/bisect_s$s.log"; grep -v "differing bins 0," "$OUT/bisect_s$s.log" | grep differing | head -5
done
for strat in partition_quads partition_chunked; do echo "== skewbench $strat"; timeout 300 python3 tools/skewbench.py --strategy $strat 2>&1 | grep -v amdgpu.ids | tee "$OUT/skew_$strat.log"; done
( timeout 1200 python -m pytest tests/test_gpu_count.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log" ); tail -5 "$OUT/pytest.log"
( KPAL_QUAD_STEPS=14 timeout 1200 python -m pytest tests/test_gpu_count.py -m gpu -x -q -k "skew or overflow or host_feed or mixed" > "$OUT/pytest_s14.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_s14.log" ); tail -5 "$OUT/pytest_s14.log"
( timeout 900 python -m pytest tests/test_gpu_vec.py tests/test_gpu_cli.py tests/test_gpu_integration_stub.py tests/test_gpu_callers.py -m gpu -x -q > "$OUT/pytest_vec.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_vec.log" ); tail -8 "$OUT/pytest_vec.log"
cd /tmp; python3 "$ROOT/bench.py" --workload matrix --steps 5 --warmup 1 > "$OUT/matrix_prod.json" 2>/dev/null; python3 -c "
import json
d=json.load(open('$OUT/matrix_prod.json')); print('matrix prod', d['ms_per_step'], d['parity_max_rel_vs_oracle_28_pairs'], d['roofline']['kernels_ms_per_step'])"
