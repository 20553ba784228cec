#!/usr/bin/env python
"""Developer harness: HIP-event times of the vector kernels (balance, pair distance, strand
balance, split) on device-resident profiles.   python tools/vbench.py [--k 12]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kpal_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument('--k', type=int, default=12)
ap.add_argument('--reps', type=int, default=5)
a = ap.parse_args()
ctx = _native.Context(0)
n = 4 ** a.k
rs = np.random.RandomState(1)
l = rs.poisson(16.6, n).astype(np.int64)
r = rs.poisson(16.6, n).astype(np.int64)
dl, dr = ctx.alloc(n * 8), ctx.alloc(n * 8)
ctx.h2d(dl, l); ctx.h2d(dr, r)
dout = ctx.alloc(n * 8)
for rep in range(a.reps + 1):
    if rep == 1:
        ctx.prof_enable(True); ctx.prof_reset()
    for metric in (0, 1, 2):
        ctx.pair_distance_device(n, dl, dr, metric)
    ctx.pair_distance_device(n, dl, dr, 0, do_balance=True, k=a.k)
    # the full option pipeline: balance + positive + smoothing (median) + scaling + cosine
    ctx.profile_distance_device(a.k, dl, dr, _native.DistanceOptions(do_balance=1, do_positive=1, do_smooth=1, summary=2,
                                                                  threshold=3.0, do_scale=1, down=0, metric=3))
    ctx.balance_device(a.k, dl)
    # summaries, merge, shrink (stat_kernels.hpp) on the device-resident vectors
    ctx.stats_device(dl, n)
    ctx._L.kpal_merge_device(ctx._h, n, dl, dr, 1, dout)
    ctx._L.kpal_shrink_device(ctx._h, a.k, 1, dl, dout)
    if a.k > 4:
        ctx._L.kpal_shrink_device(ctx._h, a.k, 4, dl, dout)
prof = ctx.prof_get()
print('k=%d  n=%d bins (%.0f MB per vector)' % (a.k, n, n * 8 / 1e6))
for name, (ms, cnt) in sorted(prof.items()):
    per = ms / cnt
    print('   %-18s %8.3f ms per launch (%d launches)' % (name, per, cnt))
print('   pair_distance reads 2 vectors: %.2f TB/s at the mean launch time' % (2 * n * 8 / (prof['pair_distance'][0] / prof['pair_distance'][1]) / 1e9))
if 'pair_distance_balanced' in prof:
    print('   pair_distance_balanced (fused): %.2f TB/s of the 2 input vectors' % (2 * n * 8 / (prof['pair_distance_balanced'][0] / prof['pair_distance_balanced'][1]) / 1e9))
for name, nbytes in (('stats', 8 * n), ('stats_var', 8 * n), ('select_hist', 8 * n), ('merge', 24 * n), ('shrink', 10 * n)):
    if name in prof:
        print('   %-12s %.2f TB/s (%d algorithmic bytes per bin)' % (name, nbytes / (prof[name][0] / prof[name][1]) / 1e9, nbytes // n))
ctx.prof_enable(False)
if a.k <= 13:
    import time
    t0 = time.perf_counter(); ctx.strand_balance(l, a.k, 0); t1 = time.perf_counter()
    f, rv = ctx.split(l, a.k); t2 = time.perf_counter()
    t3 = time.perf_counter(); st = ctx.stats(l); t4 = time.perf_counter()
    m0 = time.perf_counter(); np.median(l); l.std(); l.mean(); m1 = time.perf_counter()
    print('   host-API stats (mean, median, std...) %.1f ms incl. upload; NumPy median+std+mean on this host %.1f ms' % ((t4 - t3) * 1e3, (m1 - m0) * 1e3))
    print('   host-API strand_balance %.1f ms, split %.1f ms (include PCIe copies)' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
