#!/usr/bin/env python
"""Diagnostic: reads sharing a prefix (adapter-like), quad vs chunked pipeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from kpal_amd import _native
ctx = _native.context()
rs = np.random.RandomState(5)
L = 151
reads = (256 << 20) // L
base = np.frombuffer(b'ACGT', dtype=np.uint8)[rs.choice(4, size=reads * L)].reshape(reads, L).copy()
base[:, L - 1] = 10
d = ctx.alloc(reads * L)
def run(name, buf, strat):
    ctx.h2d(d, buf.reshape(-1))
    for it in range(3):
        if it == 1:
            ctx.prof_enable(True); ctx.prof_reset()
        ctx.count_begin(12, strat)
        ctx.count_feed_device(d, buf.size)
        ctx.count_finish(to_host=False)
    prof = ctx.prof_get(); ctx.prof_enable(False)
    print('%-44s %-18s %s' % (name, strat, ', '.join('%s %.2f' % (k, v[0] / 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:3])), flush=True)
for plen in (0, 20, 40):
    b = base.copy()
    if plen:
        b[:, :plen] = b[0, :plen]
    for strat in ('auto', 'partition_quads', 'partition_chunked'):
        run('all reads share a %d-base prefix' % plen, b, strat)
b = base.copy()
hit = rs.rand(reads) < 0.3
b[hit, :40] = b[0, :40]
for strat in ('auto', 'partition_quads', 'partition_chunked'):
    run('30 %% of the reads share a 40-base prefix', b, strat)
