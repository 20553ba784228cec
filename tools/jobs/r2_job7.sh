#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job7; mkdir -p "$OUT"
timeout 300 python3 tools/diag/quad_bisect.py > "$OUT/bisect.log" 2>&1; cat "$OUT/bisect.log"
cd /tmp && export TMPDIR=/tmp
for strat in partition_quads; do
  python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu --strategy $strat > "$OUT/bench_$strat.json" 2> "$OUT/bench_$strat.err"
  cat "$OUT/bench_$strat.json"; tail -2 "$OUT/bench_$strat.err"
done
python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --k 11 --strategy partition_quads > "$OUT/bench_k11.json" 2>&1; cat "$OUT/bench_k11.json"
cd "$ROOT"
( timeout 1200 python -m pytest tests/test_gpu_count.py -m gpu -x -q --durations=8 > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" )
tail -15 "$OUT/pytest_count.log"
