#!/usr/bin/env python
"""Developer harness: which part of low-complexity input costs the k = 12 scatter what (kernel times, 1 GiB resident in HBM):
uniform reads with a share of poly-A reads, of (AC)n reads, of both; per case the plan and the slow-path statistics.
    python tools/skewdiag.py [--share 0.02]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kpal_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument('--share', type=float, default=0.02)
ap.add_argument('--k', type=int, default=12)
a = ap.parse_args()
ctx = _native.context()
L = 151
reads = (1 << 30) // L
d = ctx.alloc(reads * L)
ctx.synth_reads_device(2, 0, reads, 150, d)
base = np.empty(reads * L, dtype=np.uint8)
ctx.d2h(base, d)
rs = np.random.RandomState(5)
hit = np.flatnonzero(rs.rand(reads) < a.share)


def variant(kind):
    m = base.reshape(reads, L).copy()
    if kind == 'polyA':
        m[hit, :150] = ord('A')
    elif kind == 'AC':
        m[hit, :150] = np.frombuffer((b'AC' * 75), dtype=np.uint8)
    elif kind == 'both':
        half = rs.rand(hit.size) < 0.5
        m[hit[half], :150] = ord('A')
        m[hit[~half], :150] = np.frombuffer((b'AC' * 75), dtype=np.uint8)
    elif kind == 'ACG':
        m[hit, :150] = np.frombuffer((b'ACG' * 50), dtype=np.uint8)
    elif kind in ('polyA_aligned', 'polyA_inside', 'polyA_inside_N'):
        # a poly-A stretch of 128 bases INSIDE the read, its flanks valid bases (polyA: the flanks are read ends) -- beginning on a
        # 16-byte boundary of the stream, or 5 bases into the read; _N: an N on either side (the flanking windows do not count)
        flat = m.reshape(-1)
        starts = hit * L + 5 if kind != 'polyA_aligned' else (hit * L + 15) // 16 * 16
        idx = (starts[:, None] + np.arange(128)[None, :]).reshape(-1)
        flat[idx] = ord('A')
        if kind == 'polyA_inside_N':
            flat[starts - 1] = ord('N')
            flat[starts + 128] = ord('N')
        return flat
    return m.reshape(-1)


for kind in ('uniform', 'polyA', 'AC', 'both', 'ACG', 'polyA_aligned', 'polyA_inside', 'polyA_inside_N'):
    buf = variant(kind)
    ctx.h2d(d, buf)
    best = None
    for it in range(4):
        before = ctx.count_stats()
        ctx.prof_enable(True); ctx.prof_reset()
        ctx.count_begin(a.k)
        ctx.count_feed_device(d, buf.size)
        plan = ctx.count_last_plan()
        ctx.count_finish(to_host=False)
        ctx.sync()
        prof = {n: v[0] for n, v in ctx.prof_get().items()}
        ctx.prof_enable(False)
        after = ctx.count_stats()
        if it and (best is None or sum(prof.values()) < sum(best.values())):
            best = prof
    st = {k2: after[k2] - before[k2] for k2 in after if after[k2] != before[k2]}
    print('%-14s %6.2f ms  %s  plan %s  %s' % (kind, sum(best.values()), ' '.join('%s %.2f' % (n, v) for n, v in sorted(best.items(), key=lambda kv: -kv[1])[:3]), plan, st), flush=True)
ctx.free(d)
