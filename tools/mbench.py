#!/usr/bin/env python
"""Developer harness for BASELINE config 5: P profiles at k, counted from synthetic reads
(seed 100+p), kdistlib.distance_matrix values on the GPU; cross-checks a few pairs against kpal_pair_distance.
    python tools/mbench.py [--P 64] [--k 12] [--reads 2000000] [--metric prod] [--balance]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kpal_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument('--P', type=int, default=64)
ap.add_argument('--k', type=int, default=12)
ap.add_argument('--reads', type=int, default=2_000_000)
ap.add_argument('--metric', default='prod')
ap.add_argument('--balance', action='store_true')
ap.add_argument('--check', type=int, default=3)
a = ap.parse_args()
ctx = _native.Context(0)
n = 4 ** a.k
nbytes = a.reads * 151
d = ctx.alloc(nbytes)
dprof = ctx.alloc(a.P * n * 8)
t0 = time.perf_counter()
host = []
for p in range(a.P):
    ctx.synth_reads_device(100 + p, 0, a.reads, 150, d)
    ctx.count_begin(a.k)
    ctx.count_feed_device(d, nbytes)
    c = ctx.count_finish()
    ctx.h2d(dprof + p * n * 8, c)
    if p < a.check + 1:
        host.append(c)
print('counted %d profiles in %.2f s' % (a.P, time.perf_counter() - t0))
metric = {'prod': 0, 'sum': 1, 'euclidean': 2}[a.metric]
ctx.distance_matrix_device(a.P, a.k, dprof, metric, a.balance)   # warm-up
ctx.prof_enable(True); ctx.prof_reset()
t0 = time.perf_counter()
vals = ctx.distance_matrix_device(a.P, a.k, dprof, metric, a.balance)
wall = time.perf_counter() - t0
for name, (ms, cnt) in sorted(ctx.prof_get().items()):
    print('   %-18s %9.3f ms (%d launches)' % (name, ms, cnt))
pairs = a.P * (a.P - 1) // 2
print('matrix P=%d k=%d metric=%s balance=%s: %.1f ms wall, %d pairs, %.1f Gterms/s, profiles read %.2f GB' % (
    a.P, a.k, a.metric, a.balance, wall * 1e3, pairs, pairs * n / wall / 1e9, a.P * n * 8 / 1e9))
# cross-check against the pair kernel (IEEE divisions, other summation order); the oracle comparisons live in tests/
worst = 0.0
for i in range(1, min(a.check + 1, a.P)):
    for j in range(i):
        want = ctx.pair_distance(host[i], host[j], metric, do_balance=a.balance, k=a.k)
        got = vals[i * (i - 1) // 2 + j]
        worst = max(worst, abs(got - want) / abs(want) if want else abs(got))
print('max rel difference vs kpal_pair_distance on %d pairs: %.3g' % (min(a.check, a.P - 1) * (min(a.check, a.P - 1) + 1) // 2, worst))
