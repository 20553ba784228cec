#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job5; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_gpu_count.py -m gpu -x -q --durations=8 > "$OUT/pytest_count.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_count.log" )
tail -25 "$OUT/pytest_count.log"
cd /tmp && export TMPDIR=/tmp
for strat in partition_quads partition_chunked; do
  python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu --strategy $strat > "$OUT/bench_$strat.json" 2> "$OUT/bench_$strat.err"
  cat "$OUT/bench_$strat.json"; tail -2 "$OUT/bench_$strat.err"
done
python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --k 11 --strategy partition_quads > "$OUT/bench_k11.json" 2>&1; cat "$OUT/bench_k11.json"
python3 "$ROOT/bench.py" --steps 5 --warmup 1 --no-cpu --k 9 --strategy partition_quads > "$OUT/bench_k9.json" 2>&1; cat "$OUT/bench_k9.json"
