#!/usr/bin/env python
"""Profile.from_sequences on a LIST of 8 million 150-base str objects (k = 12), three calls: what the host-side gatherer costs per read.
KPAL_GATHER_THREADS sets the gatherer's threads (profiles/r5/seqbench.log: one thread against sixteen).  Run on the GPU box."""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from kpal_amd import klib, _native
import oracle
n = 8_000_000
buf = oracle.synth_reads(2, 0, n, 150)
flat = bytes(buf)
t = time.perf_counter(); seqs = [flat[i * 151:i * 151 + 150].decode('ascii') for i in range(n)]; print('built %d str in %.1f s' % (n, time.perf_counter() - t))
for rep in range(3):
    t = time.perf_counter(); p = klib.Profile.from_sequences(seqs, 12); p.counts; dt = time.perf_counter() - t   # (incl. the table's download, as before round 6)
    print('threads', os.environ.get('KPAL_GATHER_THREADS', '16'), 'from_sequences %d str: %.1f ms  %.2f Gbases/s  (total %d)' % (n, dt * 1e3, n * 150 / dt / 1e9, int(p.total)))
