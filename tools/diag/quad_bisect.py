#!/usr/bin/env python
"""Diagnostic: quad pipeline vs oracle, bin for bin, at several sizes / k."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from kpal_amd import _native
ctx = _native.Context(0)
print('library:', os.environ.get('KPAL_HIP_LIBRARY', 'default'))
for k, n_reads in ((12, 200_000), (12, 400_000), (12, 700_000), (11, 700_000), (10, 700_000), (8, 700_000)):
    buf = oracle.synth_reads(41, 0, n_reads, 150, noisy=True)
    want = oracle.count_flat(buf, k, threads=8)
    d = ctx.alloc(buf.size)
    ctx.h2d(d, buf)
    for strat in ('partition_quads',):
        ctx.count_begin(k, strat)
        ctx.count_feed_device(d, buf.size)
        got = ctx.count_finish()
        diff = got - want
        nz = np.nonzero(diff)[0]
        print('k=%d reads=%d (%d tiles): %s: differing bins %d, sum(diff)=%d, sum|diff|=%d, total want %d' % (
            k, n_reads, (buf.size + 98303) // 98304, strat, nz.size, int(diff.sum()), int(np.abs(diff).sum()), int(want.sum())), flush=True)
    ctx.free(d)
