#!/usr/bin/env python
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters (one or more pass directories).

    python tools/pmc_counters.py "<note>" <dir> [<dir> ...] > out.json

Every *counter_collection.csv below the directories is read; per kernel and counter the values are
summed over the rows of a dispatch (rocprofv3 writes one row per counter instance) and averaged over
dispatches.  SQ_* cycle counters count quad-cycles summed over all SEs/XCDs (MI355X_MICROARCH.md,
rocprofv3 PMC slots); ratios between counters of one kernel are what to read."""
import collections
import csv
import json
import os
import subprocess
import sys


def main():
    note, roots = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
    for root in roots:
        for d, _, files in os.walk(root):
            for f in files:
                if not f.endswith('counter_collection.csv'):
                    continue
                with open(os.path.join(d, f)) as fh:
                    for row in csv.DictReader(fh):
                        name = row['Kernel_Name'].split('(')[0].replace('void ', '')
                        rec = acc[name][row['Counter_Name']]
                        rec[0] += float(row['Counter_Value'])
                        rec[1].add(row['Dispatch_Id'])
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_sha
    try:
        head = subprocess.check_output(['git', 'rev-parse', 'HEAD'], stderr=subprocess.DEVNULL,
                                       cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
    except Exception:
        head = os.environ.get('KPAL_HEAD', 'unknown')
    out = {'note': note, 'head': head, 'src_sha': source_sha(), 'kernels': {}}
    for name, counters in sorted(acc.items()):
        rec = {}
        for c, (total, ids) in sorted(counters.items()):
            rec[c] = total / max(len(ids), 1)
            rec['dispatches'] = len(ids)
        out['kernels'][name] = rec
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
