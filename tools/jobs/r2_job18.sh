#!/bin/bash
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r2_job18; mkdir -p "$OUT"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -fvisibility=hidden -Wno-unused-function"
( hipcc $FLAGS -DKPAL_QUAD_NT -o /tmp/lib_nt.so kpal_amd/csrc/kpal_hip.hip > "$OUT/nt_build.log" 2>&1 & )
timeout 300 python3 tools/diag/prefix.py 2>&1 | grep -v amdgpu.ids | grep "auto\|chunked" | tee "$OUT/prefix_auto.log"
for strat in auto; do echo "== skewbench $strat"; timeout 300 python3 tools/skewbench.py --strategy $strat 2>&1 | grep -v amdgpu.ids | tee "$OUT/skew_$strat.log"; done
( timeout 1700 python -m pytest tests -m gpu -x -q --durations=10 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log" ); tail -18 "$OUT/pytest.log"
cd /tmp && export TMPDIR=/tmp
wait
for lib in default /tmp/lib_nt.so default /tmp/lib_nt.so; do
  if [ "$lib" = default ]; then unset KPAL_HIP_LIBRARY; else export KPAL_HIP_LIBRARY=$lib; fi
  python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu > "$OUT/bench.json" 2> "$OUT/bench.err"
  python3 -c "
import json,sys
d=json.load(open('$OUT/bench.json'))
print('lib $lib', round(d['value'],1), round(d['ms_per_step'],3), d['checksum_ok'], {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})"
done
