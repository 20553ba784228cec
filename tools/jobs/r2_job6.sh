#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job6; mkdir -p "$OUT"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -fvisibility=hidden -Wno-unused-function"
hipcc $FLAGS -DKPAL_QUAD_NO_CARRY -o /tmp/lib_nocarry.so kpal_amd/csrc/kpal_hip.hip &
hipcc $FLAGS -DKPAL_QUAD_NO_HOT -o /tmp/lib_nohot.so kpal_amd/csrc/kpal_hip.hip &
wait
timeout 300 python3 tools/diag/quad_bisect.py > "$OUT/default.log" 2>&1; cat "$OUT/default.log"
KPAL_HIP_LIBRARY=/tmp/lib_nocarry.so timeout 300 python3 tools/diag/quad_bisect.py > "$OUT/nocarry.log" 2>&1; cat "$OUT/nocarry.log"
KPAL_HIP_LIBRARY=/tmp/lib_nohot.so timeout 300 python3 tools/diag/quad_bisect.py > "$OUT/nohot.log" 2>&1; cat "$OUT/nohot.log"
