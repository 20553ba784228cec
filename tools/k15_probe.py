#!/usr/bin/env python
"""Kernel times of the two-level quad pipeline with the plain and with the fused (balancing) finalisation
(GPU box; KPAL_HIP_LIBRARY selects a build).   python tools/k15_probe.py [k] [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kpal_amd import _native
k = int(sys.argv[1]) if len(sys.argv) > 1 else 15
reads = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
ctx = _native.Context(0)
nbytes = reads * 151
d = ctx.alloc(nbytes)
ctx.synth_reads_device(4, 0, reads, 150, d)
for mode in ('plain', 'balanced'):
    for it in range(4):
        if it == 1:
            ctx.prof_enable(True)
            ctx.prof_reset()
        ctx.count_begin(k)
        ctx.count_feed_device(d, nbytes)
        if mode == 'balanced':
            ctx.count_balance()
        ctx.count_finish(to_host=False)
        ctx.sync()
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    print('%-9s' % mode, '  '.join('%s %.3f' % (n, v[0] / v[1]) for n, v in sorted(prof.items()) if v[1]), ' plan', ctx.count_last_plan(), flush=True)
ctx.free(d)
ctx.close()
