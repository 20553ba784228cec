#!/usr/bin/env python
"""Randomised parity stress of the vector side (developer tool, GPU box): balance, split, strand
balance, summaries / merge / shrink, every ProfileDistance option combination and the distance matrix
against the oracle.
    python tests/stress_vec.py [--seconds 90] [--seed 1]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kpal_amd import _native
import oracle

ap = argparse.ArgumentParser()
ap.add_argument('--seconds', type=float, default=90)
ap.add_argument('--seed', type=int, default=1)
a = ap.parse_args()
rs = np.random.RandomState(a.seed)
ctx = _native.context()
code = {'min': 0, 'average': 1, 'median': 2}
metric = {'prod': 0, 'sum': 1, 'euclidean': 2, 'cosine': 3}


def close(x, y):
    if np.isnan(y):
        return np.isnan(x)
    if np.isinf(y):
        return x == y
    # (wrapping int64 inputs give terms of both signs: a sum that cancels to ~1e-16 is only good to ~1e-16 absolute)
    return abs(x - y) <= 1e-9 * abs(y) + 1e-12


def vector(n):
    kind = rs.randint(5)
    if kind == 0:
        v = rs.poisson(rs.choice([0.3, 3, 800]), n)
    elif kind == 1:
        v = rs.randint(0, 1 << 40, n)
    elif kind == 2:
        v = rs.randint(1 << 31, 1 << 33, n)            # beyond the float fast path of the matrix kernel
    elif kind == 3:
        v = rs.poisson(5, n) * (rs.rand(n) < 0.1)
    else:
        v = rs.randint(0, 1 << 61, n)
    return v.astype(np.int64)


t_end = time.time() + a.seconds
cases = 0
while time.time() < t_end:
    k = int(rs.randint(1, 11))
    n = 4 ** k
    l, r = vector(n), vector(n)
    b = l.copy()
    ctx.balance_inplace(b, k)
    assert np.array_equal(b, oracle.balance(l, k)), ('balance', k)
    f, rv = ctx.split(l, k)
    fo, ro = oracle.split(l, k)
    assert np.array_equal(f, fo) and np.array_equal(rv, ro), ('split', k)
    for pw in ('prod', 'sum'):
        assert close(ctx.strand_balance(l, k, metric[pw]), oracle.strand_balance(l, k, pw)), ('strand', k, pw)
    # summaries, merge, shrink (stat_kernels.hpp)
    sv = l if rs.rand() < 0.7 else (l - rs.randint(0, 1 << 20)).astype(np.int64)      # sometimes negative entries
    got, want = ctx.stats(sv), oracle.stats(sv)
    assert (got.total, got.non_zero, got.min, got.max, got.median) == (want['total'], want['non_zero'], want['min'], want['max'], want['median']), ('stats', k)
    scale = float(np.abs(sv.astype(np.float64)).mean())
    assert abs(got.mean - want['mean']) <= 1e-9 * scale + 1e-300 and abs(got.std - want['std']) <= 1e-9 * abs(want['std']) + 1e-300, ('stats fp', k, got.mean, want['mean'], got.std, want['std'])
    mname = ('sum', 'xor', 'int', 'nint')[rs.randint(4)]
    assert np.array_equal(ctx.merge(l, r, ('sum', 'xor', 'int', 'nint').index(mname)), oracle.merge(l, r, mname)), ('merge', k, mname)
    if k > 1:
        f_ = int(rs.randint(1, k))
        assert np.array_equal(ctx.shrink(l, k, f_), oracle.shrink(l, k, f_)), ('shrink', k, f_)
    for _ in range(6):
        o = dict(do_balance=bool(rs.rand() < 0.5), do_positive=bool(rs.rand() < 0.3), do_smooth=bool(rs.rand() < 0.5),
                 summary=['min', 'average', 'median'][rs.randint(3)], threshold=[0, 1, 2.5, 100][rs.randint(4)],
                 do_scale=bool(rs.rand() < 0.4), down=bool(rs.rand() < 0.5),
                 metric=['prod', 'sum', 'euclidean', 'cosine'][rs.randint(4)])
        opt = _native.DistanceOptions(do_balance=o['do_balance'], do_positive=o['do_positive'], do_smooth=o['do_smooth'],
                                      summary=code[o['summary']], threshold=o['threshold'], do_scale=o['do_scale'],
                                      down=o['down'], metric=metric[o['metric']])
        with np.errstate(all='ignore'):
            e = oracle.profile_distance(l, r, k, **o)
        v = ctx.profile_distance(l, r, k, opt)
        assert close(v, e), (k, o, v, e)
    if k <= 8:
        P = int(rs.randint(2, 7))
        profs = [vector(n) for _ in range(P)]
        for m in ('prod', 'sum', 'euclidean'):
            bal = bool(rs.rand() < 0.5)
            got = ctx.distance_matrix(profs, k, metric[m], do_balance=bal)
            want = oracle.distance_matrix_values(profs, k, do_balance=bal, metric=m)
            assert all(close(x, y) for x, y in zip(got, want)), ('matrix', k, m, bal)
    if 6 <= k <= 8 and rs.rand() < 0.25:
        # 9 .. 70 profiles: the super-tile kernels (9..16, > 64) and the kernels that stage every profile once (17..64; 256 and 1024
        # threads), on count-like vectors -- Poisson of a random mean with zero bins, a few counts past the reciprocal tables, an
        # empty profile now and then
        P = int(rs.randint(9, 71))
        mean = float(rs.choice([0.3, 4.0, 60.0, 700.0]))
        profs = [rs.poisson(mean, n).astype(np.int64) for _ in range(P)]
        for pr in profs[::5]:
            pr[rs.randint(0, n, 20)] = rs.randint(0, 3000, 20)
        if rs.rand() < 0.3:
            profs[int(rs.randint(P))][:] = 0
        for m in ('prod', 'sum'):
            bal = bool(rs.rand() < 0.3)
            got = ctx.distance_matrix(profs, k, metric[m], do_balance=bal)
            want = oracle.distance_matrix_values(profs, k, do_balance=bal, metric=m)
            assert all(close(x, y) for x, y in zip(got, want)), ('matrix', k, P, m, bal)
    cases += 1
print('vector stress ok: %d cases, seed %d' % (cases, a.seed))
