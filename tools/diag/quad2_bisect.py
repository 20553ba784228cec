#!/usr/bin/env python
"""Diagnostic: two-level quad pipeline (k = 13..16) vs oracle / global atomics, bin for bin."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import oracle
from kpal_amd import _native, dist
ctx = _native.Context(0)
other = _native.Context(0)
for k, n_reads, noisy in ((13, 300_000, True), (14, 700_000, True), (15, 700_000, False), (15, 2_000_000, True), (16, 700_000, True)):
    buf = oracle.synth_reads(41, 0, n_reads, 150, noisy=noisy)
    d = ctx.alloc(buf.size)
    ctx.h2d(d, buf)
    ctx.count_begin(k, 'partition2_quads')
    ctx.count_feed_device(d, buf.size)
    ctx.count_finish(to_host=False)
    ctx.sync()
    a = dist.table_as_tensor(ctx)
    other.count_begin(k, 'global_atomic')
    other.count_feed_device(d, buf.size)
    other.count_finish(to_host=False)
    other.sync()
    b = dist.table_as_tensor(other)
    diff = a - b
    nd = int((diff != 0).sum())
    print('k=%d reads=%d: differing bins %d, sum(diff)=%d, sum|diff|=%d, total %d (want %d)' % (
        k, n_reads, nd, int(diff.sum()), int(diff.abs().sum()), int(a.sum()), int(b.sum())), flush=True)
    if nd:
        idx = torch.nonzero(diff).flatten()[:6]
        print('   samples:', [(hex(int(i)), int(a[i]), int(b[i])) for i in idx])
    torch.cuda.synchronize()
    ctx.free(d)
