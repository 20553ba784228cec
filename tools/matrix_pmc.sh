set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3w; mkdir -p $OUT; B=$ROOT/bench.py
M5="--workload matrix --steps 2 --warmup 1 --no-cpu"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch_matrix" -o p -- python3 "$B" $M5 > /dev/null 2> "$OUT/pmc_fetch_matrix.err"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write_matrix" -o p -- python3 "$B" $M5 > /dev/null 2> "$OUT/pmc_write_matrix.err"
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_fetch_matrix" "$OUT/pmc_write_matrix" 0 "matrix" matrix_rdiff > "$OUT/pmc_hbm_traffic_matrix.json"
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d "$OUT/pmc_l2_matrix" -o p -- python3 "$B" $M5 > /dev/null 2> "$OUT/pmc_l2_matrix.err"
tail -3 "$OUT/pmc_l2_matrix.err"
python3 - "$OUT/pmc_l2_matrix" <<'PY'
import collections, csv, json, os, sys
per = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for d, _, files in os.walk(sys.argv[1]):
    for f in files:
        if f.endswith('counter_collection.csv'):
            for row in csv.DictReader(open(os.path.join(d, f))):
                name = row['Kernel_Name'].split('(')[0].replace('void ', '')
                per[name][row['Counter_Name']] += float(row['Counter_Value']); disp[name].add(row['Dispatch_Id'])
print(json.dumps({n: dict({k: v / len(disp[n]) for k, v in c.items()}, dispatches=len(disp[n])) for n, c in per.items() if 'matrix' in n}, indent=1))
PY
python3 -c "
import json; d=json.load(open('$OUT/pmc_hbm_traffic_matrix.json'))
for k,v in d['kernels'].items():
    if 'matrix' in k or 'reduce' in k: print(k, v)
"
find $OUT -name '*.db' -delete; find $OUT -name '*counter_collection.csv' -delete
