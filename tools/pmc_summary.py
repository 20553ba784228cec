#!/usr/bin/env python
"""Summarise two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs of the
same command) into per-kernel HBM bytes per dispatch.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <input_bytes_of_the_run> "<note>" <kernel> > out.json

<input_bytes_of_the_run> = bytes fed over all profiled steps (warm-up included); it is divided by the number of
dispatches of <kernel> (name substring) to give the average input bytes per launch that bench.py scales by.

FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section), so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def collect(root, counter):
    per = collections.defaultdict(lambda: [0.0, set()])
    for d, _, files in os.walk(root):
        for f in files:
            if not f.endswith('counter_collection.csv'):
                continue
            with open(os.path.join(d, f)) as fh:
                for row in csv.DictReader(fh):
                    if row.get('Counter_Name') != counter:
                        continue
                    name = row['Kernel_Name'].split('(')[0].replace('void ', '')
                    per[name][0] += float(row['Counter_Value'])
                    per[name][1].add(row['Dispatch_Id'])
    return {k: (v[0], len(v[1])) for k, v in per.items()}


def main():
    fetch_dir, write_dir, run_bytes, note, kernel = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
    fetch = collect(fetch_dir, 'FETCH_SIZE')
    write = collect(write_dir, 'WRITE_SIZE')
    launches = [n for name, (_, n) in fetch.items() if kernel in name]
    if len(launches) != 1:
        sys.exit('pmc_summary: %d kernels match %r' % (len(launches), kernel))
    from bench import source_sha
    out = {'note': note, 'head': os.environ.get('KPAL_HEAD', 'unknown'), 'src_sha': source_sha(), 'input_bytes_of_the_run': run_bytes, 'launches_of': kernel, 'launches': launches[0],
           'input_bytes_per_launch_avg': run_bytes / launches[0], 'kernels': {}}
    for name in fetch:
        f, n = fetch[name]
        w, nw = write.get(name, (0.0, n))
        rec = {'FETCH_SIZE': f / n, 'dispatches': n, 'WRITE_SIZE': w / max(nw, 1)}
        rec['hbm_bytes_per_dispatch_corrected'] = (2 * rec['FETCH_SIZE'] + rec['WRITE_SIZE']) * 1024
        out['kernels'][name] = rec
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
