#!/usr/bin/env python
"""Developer micro-harness: per-kernel HIP-event times of one counting configuration.
    python tools/kbench.py [--k 12] [--reads 20000000] [--strategy auto] [--steps 3]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kpal_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument('--k', type=int, default=12)
ap.add_argument('--reads', type=int, default=20_000_000)
ap.add_argument('--strategy', default='auto')
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--check', action='store_true')
a = ap.parse_args()
ctx = _native.Context(0)
nbytes = a.reads * 151
d = ctx.alloc(nbytes)
ctx.synth_reads_device(2, 0, a.reads, 150, d)
for it in range(a.steps + 1):
    if it == 1:
        ctx.prof_enable(True); ctx.prof_reset()
    ctx.count_begin(a.k, a.strategy)
    ctx.count_feed_device(d, nbytes)
    ctx.count_finish(to_host=False)
prof = ctx.prof_get()
tot = sum(v[0] for v in prof.values()) / a.steps
print('k=%d reads=%d strategy=%s ablate=%s: %.3f ms/step kernels, %.1f Gbases/s' % (
    a.k, a.reads, a.strategy, os.environ.get('KPAL_ABLATE', '0'), tot, a.reads * 150 / tot / 1e6))
for n, (ms, cnt) in sorted(prof.items()):
    print('   %-22s %8.3f ms/step  (%d launches/step, %.3f ms each)' % (n, ms / a.steps, cnt // a.steps, ms / cnt))
if a.check:
    out = ctx.count_finish()
    print('   total', int(out.sum()), 'expected', a.reads * (150 - a.k + 1))
