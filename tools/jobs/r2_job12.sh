#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job12; mkdir -p "$OUT"
for s in 0 13 8 6 4 3 2; do
  KPAL_QUAD_STEPS=$s timeout 300 python3 tools/diag/quad_bisect.py > "$OUT/bisect_s$s.log" 2>&1; echo "steps $s: ok lines $(grep -c 'differing bins 0,' $OUT/bisect_s$s.log)"; grep -v "differing bins 0," "$OUT/bisect_s$s.log" | grep differing | head -5
done
for strat in partition_quads; do echo "== skewbench $strat"; timeout 300 python3 tools/skewbench.py --strategy $strat 2>&1 | grep -v amdgpu.ids | tee "$OUT/skew_$strat.log"; done
cd /tmp && export TMPDIR=/tmp
for s in 0 12 13 0; do
  KPAL_QUAD_STEPS=$s python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu > "$OUT/bench_s$s.json" 2> "$OUT/bench.err"
  python3 -c "
import json,sys
d=json.load(open('$OUT/bench_s$s.json'))
print('steps $s', round(d['value'],1), round(d['ms_per_step'],3), d['checksum_ok'], {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
cd "$ROOT"
( timeout 1200 python -m pytest tests/test_gpu_count.py tests/test_gpu_integration_stub.py tests/test_gpu_cli.py tests/test_gpu_callers.py tests/test_gpu_fasta.py -m gpu -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log" ); tail -8 "$OUT/pytest.log"
for s in 8 3; do ( KPAL_QUAD_STEPS=$s timeout 600 python -m pytest tests/test_gpu_count.py -m gpu -x -q -k "skew or overflow or host_feed or mixed or g2 or g3" > "$OUT/pytest_s$s.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_s$s.log" ); tail -3 "$OUT/pytest_s$s.log"; done
