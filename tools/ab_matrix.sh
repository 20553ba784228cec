#!/bin/bash
# On the GPU box: the multiset matrix kernels of every build/variants/lib_*.so on BASELINE config 5 (tools/mbench.py).
#   bash tools/ab_matrix.sh OUT [mbench arguments]
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for lib in build/variants/lib_*.so; do
  n=$(basename "$lib" .so)
  echo "== ${n#lib_}" >> "$OUT"
  KPAL_HIP_LIBRARY=$lib timeout 300 python3 tools/mbench.py --check 0 "$@" 2>&1 | grep -E "matrix_|FAILED|Error" >> "$OUT"
done
cat "$OUT"
