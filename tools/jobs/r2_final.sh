#!/bin/bash
# End-of-round run on the GPU box: the whole GPU suite, the smoke entry, then the round's profile recipe.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_final; mkdir -p "$OUT"
( timeout 1800 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log" ); tail -4 "$OUT/pytest_gpu.log"
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" >> "$OUT/smoke.log" ); tail -2 "$OUT/smoke.log"
bash tools/jobs/r2_job33.sh
