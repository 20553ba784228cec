#!/usr/bin/env python
"""Diagnostic: k=15 two-level path vs the global-atomic kernel at growing batch sizes (on the device)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from kpal_amd import _native, dist

k = 15
ctx = _native.Context(0)
other = _native.Context(0)
sizes = [int(x) for x in sys.argv[1:]] or [20_000_000, 30_000_000, 50_000_000, 100_000_000]
nmax = max(sizes)
d = ctx.alloc(nmax * 151)
ctx.synth_reads_device(4, 0, nmax, 150, d)
for n in sizes:
    ctx.count_begin(k)
    ctx.count_feed_device(d, n * 151)
    ctx.count_finish(to_host=False)
    ctx.sync()
    a = dist.table_as_tensor(ctx)
    other.count_begin(k, 'global_atomic')
    other.count_feed_device(d, n * 151)
    other.count_finish(to_host=False)
    other.sync()
    b = dist.table_as_tensor(other)
    diff = (a != b)
    nd = int(diff.sum())
    print('n=%d: total a=%d b=%d want=%d, differing bins=%d' % (n, int(a.sum()), int(b.sum()), n * (150 - k + 1), nd), flush=True)
    if nd:
        idx = torch.nonzero(diff).flatten()
        print('   first/last differing bin: %d .. %d; coarse buckets hit: %s' % (int(idx[0]), int(idx[-1]), sorted(set((idx >> 24).cpu().numpy().tolist()))[:70]))
        print('   sum(a-b) over differing = %d; sum|a-b| = %d' % (int((a - b)[diff].sum()), int((a - b)[diff].abs().sum())))
        sub = idx[:8]
        print('   samples:', [(int(i), int(a[i]), int(b[i])) for i in sub])
        fine = ((idx >> 15) & 511)
        print('   fine buckets hit (count):', int(torch.unique(fine).numel()), ' low15 range', int((idx & 32767).min()), int((idx & 32767).max()))
