#!/usr/bin/env python
"""Developer harness: PCIe-inclusive rates of the host-facing entry points (never the bench value):
kpal_count_feed on a host buffer, Profile.from_sequences on a list of reads, Profile.from_fasta on
FASTA text, and count_finish's D2H of the table.   python tools/hostbench.py [--k 12] [--reads 4000000]"""
import argparse, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kpal_amd import _native, klib

ap = argparse.ArgumentParser()
ap.add_argument('--k', type=int, default=12)
ap.add_argument('--reads', type=int, default=4_000_000)
a = ap.parse_args()
ctx = _native.context()
nbytes = a.reads * 151
d = ctx.alloc(nbytes)
ctx.synth_reads_device(2, 0, a.reads, 150, d)
host = np.empty(nbytes, dtype=np.uint8)
ctx.d2h(host, d)
bases = a.reads * 150


def timed(label, fn, reps=3):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print('   %-46s %8.1f ms  %7.2f Gbases/s' % (label, best * 1e3, bases / best / 1e9))
    return out


def feed():
    ctx.count_begin(a.k)
    ctx.count_feed(host)
    return ctx.count_finish(to_host=False)


def feed_d2h():
    ctx.count_begin(a.k)
    ctx.count_feed(host)
    return ctx.count_finish()


print('k=%d, %d reads x 150 bp (%.2f GB host buffer); from_sequences gatherer: %s' % (a.k, a.reads, nbytes / 1e9, 'C extension' if klib._kpal_gather else 'interpreter join'))
timed('kpal_count_feed (host buffer, table stays on GPU)', feed)
ref = timed('kpal_count_feed + D2H of the 4^k table', feed_d2h)
reads = [bytes(r) for r in host.reshape(-1, 151)[:min(a.reads, 1_000_000), :150]]
scale = len(reads) / a.reads
_b = bases
bases = int(bases * scale)
def touched(profile):
    profile.counts          # (since round 6 the table stays in HBM until asked for: the download belongs to these figures)
    return profile


p = timed('Profile.from_sequences (%d bytes objects)' % len(reads), lambda: touched(klib.Profile.from_sequences(reads, a.k)))
sreads = [r.decode() for r in reads]
timed('Profile.from_sequences (%d str objects)' % len(reads), lambda: touched(klib.Profile.from_sequences(sreads, a.k)))
timed('   ... the table left in HBM (no D2H)', lambda: klib.Profile.from_sequences(sreads, a.k))
bases = _b
fa = b''.join(b'>r%d\n' % i + r[:75] + b'\n' + r[75:] + b'\n' for i, r in enumerate(reads))
def fasta_native():
    ctx.count_begin(a.k)
    ctx.count_feed_fasta(fa)
    return ctx.count_finish(to_host=False)


_b = bases
bases = int(bases * scale)
timed('kpal_count_feed_fasta (same text, one call)', fasta_native)
bases = _b
bases = int(bases * scale)
q = timed('Profile.from_fasta (%.0f MB of text, 2 lines/record)' % (len(fa) / 1e6), lambda: touched(klib.Profile.from_fasta(io.BytesIO(fa), a.k)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); klib.Profile.from_fasta(io.BytesIO(fa), a.k); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
assert (p.counts == q.counts).all()
if len(reads) == a.reads:
    assert (p.counts == ref).all()
