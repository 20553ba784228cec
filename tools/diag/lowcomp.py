#!/usr/bin/env python
"""Diagnostic: quad scatter time on reads with a fraction of low-complexity reads."""
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
def run(name, buf):
    ctx.h2d(d, buf.reshape(-1))
    for it in range(3):
        if it == 1:
            ctx.prof_enable(True); ctx.prof_reset()
        ctx.count_begin(12, 'partition_quads')
        ctx.count_feed_device(d, buf.size)
        ctx.count_finish(to_host=False)
    prof = ctx.prof_get(); ctx.prof_enable(False)
    print('%-40s scatter %.2f ms hist %.2f ms' % (name, prof['quad_scatter'][0] / 2, prof['quad_hist'][0] / 2), flush=True)
run('uniform', base)
for frac in (0.002, 0.02):
    for unit, nm in ((b'A', 'polyA'), (b'AC', '(AC)n'), (b'ACGTTGCA', '(ACGTTGCA)n'), (b'AACCGGTTAGCATCGA', 'period 16')):
        b = base.copy()
        hit = rs.rand(reads) < frac
        b[hit, :150] = np.resize(np.frombuffer(unit, dtype=np.uint8), 150)
        run('%.1f %% %s reads' % (100 * frac, nm), b)
# every read starts with the same 40 bases (adapter-like)
b = base.copy(); b[:, :40] = b[0, :40]
run('all reads share a 40-base prefix', b)
