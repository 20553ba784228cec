#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job3; mkdir -p "$OUT"
hipcc -O3 --offload-arch=gfx950 -o /tmp/tlb_probe tools/tlb_probe.hip && timeout 600 /tmp/tlb_probe > "$OUT/tlb_probe.log" 2>&1
cat "$OUT/tlb_probe.log"
( timeout 1500 python -m pytest tests -m gpu -q --durations=12 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log" )
tail -40 "$OUT/pytest.log"
