#!/usr/bin/env python
"""Developer harness: counting rate on non-uniform inputs (kernel time, input resident in HBM):
AT-rich reads, reads with low-complexity stretches, and an assembled-genome-like single record with
slowly drifting composition.   python tools/skewbench.py [--k 12] [--mb 1024]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kpal_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument('--k', type=int, default=12)
ap.add_argument('--mb', type=int, default=1024)
ap.add_argument('--strategy', default='auto')
a = ap.parse_args()
ctx = _native.context()
rs = np.random.RandomState(5)
n = a.mb << 20
L = 151
reads = n // L


def sample(p, size):
    return np.frombuffer(b'ACGT', dtype=np.uint8)[rs.choice(4, size=size, p=p)]


def as_reads(flat):
    m = flat[:reads * L].reshape(reads, L).copy()
    m[:, L - 1] = 10
    return m.reshape(-1)


cases = {}
cases['uniform'] = as_reads(sample([.25, .25, .25, .25], reads * L))
cases['AT-rich (A,T 0.32; C,G 0.18)'] = as_reads(sample([.32, .18, .18, .32], reads * L))
low = as_reads(sample([.25, .25, .25, .25], reads * L)).reshape(reads, L)
hit = rs.rand(reads) < 0.02                      # 2 % of the reads are poly-A / (AC)n
low[hit, :150] = np.where(rs.rand(hit.sum(), 1) < 0.5, ord('A'), np.tile(np.frombuffer(b'AC', dtype=np.uint8), 75))
cases['2 % low-complexity reads'] = low.reshape(-1)
# one long record, GC content drifting between 35 % and 55 % over ~2 MB windows
win = 1 << 21
gc = 0.45 + 0.10 * np.sin(np.arange((reads * L + win - 1) // win) * 0.7)
parts = [sample([(1 - g) / 2, g / 2, g / 2, (1 - g) / 2], win) for g in gc]
cases['one record, drifting GC'] = np.concatenate(parts)[:reads * L]
cases['homopolymer (all A)'] = np.full(reads * L, ord('A'), dtype=np.uint8)

d = ctx.alloc(reads * L)
for name, buf in cases.items():
    ctx.h2d(d, buf)
    for it in range(3):
        if it == 1:
            ctx.prof_enable(True); ctx.prof_reset()
        ctx.count_begin(a.k, a.strategy)
        ctx.count_feed_device(d, buf.size)
        ctx.count_finish(to_host=False)
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    ms = sum(v[0] for v in prof.values()) / 2
    top = ', '.join('%s %.2f' % (k2, v[0] / 2) for k2, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:3])
    print('%-34s %8.2f ms  %7.1f Gbases/s   (%s)' % (name, ms, buf.size * 150 / 151 / ms / 1e6, top))
