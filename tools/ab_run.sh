#!/bin/bash
# On the GPU box: time every build/variants/lib_*.so (tools/ab_build.py) and the regular build on one bench command.
#   bash tools/ab_run.sh OUTDIR [bench arguments ...]      (default: --reads 20000000 --steps 5 --warmup 1)
OUT=$1; shift
ARGS=${*:---reads 20000000 --steps 5 --warmup 1}
mkdir -p "$OUT"
run() {   # name, library
  KPAL_HIP_LIBRARY=$2 python3 bench.py --no-cpu --no-extra $ARGS > "$OUT/$1.json" 2> "$OUT/$1.err"
  python3 - "$1" "$OUT/$1.json" <<'PY'
import json, sys
try:
    line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    ks = line['roofline']['kernels_ms_per_step']
    print('%-22s %7.3f ms/step  ' % (sys.argv[1], line['ms_per_step']) + '  '.join('%s %.3f' % (k, v) for k, v in sorted(ks.items())))
except Exception as e:
    print('%-22s FAILED %s' % (sys.argv[1], e))
PY
}
run regular kpal_amd/libkpal_hip.so
for lib in build/variants/lib_*.so; do
  n=$(basename "$lib" .so); run "${n#lib_}" "$lib"
done
