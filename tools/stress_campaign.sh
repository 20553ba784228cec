#!/bin/bash
# On the GPU box: the randomised parity stress tools on a range of seeds, one summary line per run.
#   bash tools/stress_campaign.sh OUT FIRST_SEED COUNT_RUNS VEC_RUNS
OUT=$1; S=$2; NC=$3; NV=$4
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
python3 - <<'PY' >> "$OUT"
import sys; sys.path.insert(0, '.')
import bench; print('src_sha', bench.source_sha())
PY
for i in $(seq 0 $((NC - 1))); do timeout 400 python3 tests/stress_count.py --seed $((S + i)) 2>&1 | tail -1 >> "$OUT"; done
for i in $(seq 0 $((NV - 1))); do timeout 300 python3 tests/stress_vec.py --seed $((S + i)) 2>&1 | tail -1 >> "$OUT"; done
cat "$OUT"
