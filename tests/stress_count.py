#!/usr/bin/env python
"""Randomised parity stress (developer tool, GPU box): random k, sizes (log-uniform 1 B .. 96 MB),
compositions, separator densities, low-complexity stretches, feed kinds and strategies against the
oracle.   python tests/stress_count.py [--seconds 150] [--seed 1]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kpal_amd import _native
import oracle

ap = argparse.ArgumentParser()
ap.add_argument('--seconds', type=float, default=150)
ap.add_argument('--seed', type=int, default=1)
a = ap.parse_args()
rs = np.random.RandomState(a.seed)
ctx = _native.context()
acgt = np.frombuffer(b'ACGTacgt', dtype=np.uint8)
t_end = time.time() + a.seconds
n_cases = 0
while time.time() < t_end:
    k = int(rs.choice([1, 3, 7, 8, 9, 10, 11, 12, 12, 12, 13, 13, 14]))
    n = int(np.exp(rs.uniform(0, np.log(96 << 20))))
    p = rs.dirichlet(np.ones(4) * rs.choice([0.3, 1.0, 5.0]))
    p8 = np.concatenate([p * 0.97, p * 0.03])
    buf = acgt[rs.choice(8, size=n, p=p8 / p8.sum())].copy()
    sep = rs.choice([0.0, 1e-4, 1e-2, 0.2])
    if sep:
        buf[rs.rand(n) < sep] = rs.choice([10, ord('N'), ord('-'), 0, 255])
    for _ in range(rs.randint(0, 4)):
        if n > 100:
            start = rs.randint(0, n - 50)
            length = min(n - start, int(np.exp(rs.uniform(np.log(50), np.log(4 << 20)))))
            unit = [b'A', b'T', b'AC', b'CAG', b'ACGTT', b'N'][rs.randint(6)]
            buf[start:start + length] = np.resize(np.frombuffer(unit, dtype=np.uint8), length)
    want = oracle.count_flat(buf, k, threads=16)
    strategies = ['auto']
    if 8 <= k <= 12:
        strategies += ['partition', 'partition_chunked', 'partition_quads']
    if k >= 13:
        strategies += ['partition2', 'partition2_quads']
    for strat in strategies:
        bal = rs.rand() < 0.4               # count + balance (k >= 13 on the quad pipeline: fused into the finalisation of the table)
        if rs.rand() < 0.5:
            ctx.count_begin(k, strat)
            ctx.count_feed(buf)
            if bal:
                ctx.count_balance()
            got = ctx.count_finish()
            if bal:
                if not np.array_equal(got, oracle.balance(want, k)):
                    print('MISMATCH k=%d n=%d strat=%s seed=%d case=%d (count + balance)' % (k, n, strat, a.seed, n_cases)); sys.exit(1)
                continue
        else:
            off = int(rs.randint(0, 16))
            d = ctx.alloc(n + 64)
            ctx.h2d(d + off, buf)
            ctx.count_begin(k, strat)
            cut = int(rs.randint(0, n + 1))
            ctx.count_feed_device(d + off, n)      # one feed ...
            if cut and rs.rand() < 0.3:            # ... or the same data again in two feeds (linearity)
                ctx.count_feed_device(d + off, cut)
                ctx.count_feed_device(d + off + cut, n - cut)
                want2 = want + oracle.count_flat(buf[:cut], k) + oracle.count_flat(buf[cut:], k)
            else:
                want2 = want
            if bal:
                ctx.count_balance()
                want2 = oracle.balance(want2, k)
            got = ctx.count_finish()
            ctx.free(d)
            if not np.array_equal(got, want2):
                print('MISMATCH k=%d n=%d strat=%s seed=%d case=%d (device feed)' % (k, n, strat, a.seed, n_cases)); sys.exit(1)
            continue
        if not np.array_equal(got, want):
            print('MISMATCH k=%d n=%d strat=%s seed=%d case=%d' % (k, n, strat, a.seed, n_cases)); sys.exit(1)
    n_cases += 1
print('stress ok: %d cases, seed %d' % (n_cases, a.seed))
