"""Parity of the ProfileDistance option pipeline on the GPU (positive, dynamic smoothing, scaling,
cosine: kpal/kdistlib.py:126-161) with the reference goldens (G9, G6) and the oracle.
Smoothed vectors bit-exact; distances within 1e-9 relative.  Run on the GPU box: pytest -m gpu."""
import io
import math

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def close(a, b, rtol=RTOL):
    if np.isnan(b):
        return np.isnan(a)
    if np.isinf(b):
        return a == b
    return abs(a - b) <= rtol * abs(b) + 1e-300


def make_distance(c):
    from kpal_amd import kdistlib, metrics
    fn = {'prod': None, 'sum': None, 'euclidean': metrics.vector_distance['euclidean'],
          'cosine': metrics.vector_distance['cosine']}[c['metric']]
    return kdistlib.ProfileDistance(
        do_balance=c['do_balance'], do_positive=c['do_positive'], do_smooth=c['do_smooth'],
        summary=metrics.summary[c['summary']], threshold=c['threshold'], do_scale=c['do_scale'], down=c['down'],
        distance_function=fn, pairwise=metrics.pairwise[c['metric'] if c['metric'] in ('prod', 'sum') else 'prod'])


def test_g9_option_grid_through_the_python_api(golden_options):
    """All 672 reference cases through kpal_amd.kdistlib.ProfileDistance (-> kpal_profile_distance)."""
    from kpal_amd import klib
    g, z = golden_options
    for c in g['cases']:
        l, r = z['g9_%d_l' % c['pair']], z['g9_%d_r' % c['pair']]
        left, right = klib.Profile(l.copy(), 'l'), klib.Profile(r.copy(), 'r')
        d = make_distance(c)
        assert d._native_options() is not None
        v = d.distance(left, right)
        assert close(v, c['distance']), (c, v)
        np.testing.assert_array_equal(left.counts, l)     # inputs are left unmodified
        np.testing.assert_array_equal(right.counts, r)


def test_g9_smoothed_vectors_bit_exact(golden_options):
    from kpal_amd import kdistlib, klib, metrics
    g, z = golden_options
    checked = 0
    for pi in range(g['n_pairs']):
        for name, fn, th in g['smoothed']:
            key = 'g9_%d_%s_l' % (pi, name)
            if key not in z:
                continue
            a = klib.Profile(z['g9_%d_l' % pi].copy())
            b = klib.Profile(z['g9_%d_r' % pi].copy())
            kdistlib.ProfileDistance(do_smooth=True, summary=metrics.summary[fn], threshold=th).dynamic_smooth(a, b)
            np.testing.assert_array_equal(a.counts, z[key])
            np.testing.assert_array_equal(b.counts, z['g9_%d_%s_r' % (pi, name)])
            checked += 1
    assert checked >= 20


def test_g6_option_known_answers(golden_scalars):
    """tests/test_kmer.py:384-405 known answers (0.077 / 0.474) and friends, k=8."""
    from kpal_amd import kdistlib, klib, metrics
    g = golden_scalars['G6']['left_right_k8']
    sets = golden_scalars['G8']['sets']
    left = klib.Profile.from_sequences(sets[0], 8)
    right = klib.Profile.from_sequences(sets[1], 8)
    D = kdistlib.ProfileDistance
    assert close(D(do_smooth=True).distance(left, right), g['smooth_min'])
    assert close(D(do_smooth=True, summary=np.mean).distance(left, right), g['smooth_avg'])
    assert close(D(do_positive=True).distance(left, right), g['positive'])
    assert close(D(do_scale=True).distance(left, right), g['scale'])
    assert close(D(do_scale=True, down=True).distance(left, right), g['scale_down'])
    assert close(D(distance_function=metrics.cosine_similarity).distance(left, right), g['cosine'])
    assert '%.3f' % D(do_smooth=True).distance(left, right) == '0.077'
    assert '%.3f' % D(do_smooth=True, summary=np.mean).distance(left, right) == '0.474'


def test_options_vs_oracle_larger_k():
    """k = 8..11 (tiled balance, many smoothing levels) against the oracle; smoothing bit-exact."""
    from kpal_amd import _native
    ctx = _native.context()
    rs = np.random.RandomState(11)
    code = {'min': 0, 'average': 1, 'median': 2}
    metric = {'prod': 0, 'sum': 1, 'euclidean': 2, 'cosine': 3}
    for k in (8, 9, 10, 11):
        n = 4 ** k
        l = rs.poisson(1.2, n).astype(np.int64)
        r = rs.poisson(0.9, n).astype(np.int64)
        l[n // 3: n // 2] = 0
        r[n // 3: n // 3 + n // 8] //= 2
        for summary, th in (('min', 0), ('average', 1.5), ('median', 1)):
            a, b = l.copy(), r.copy()
            ctx.dynamic_smooth(a, b, k, code[summary], th)
            ao, bo = oracle.dynamic_smooth(l, r, k, summary, th)
            np.testing.assert_array_equal(a, ao)
            np.testing.assert_array_equal(b, bo)
            assert a.sum() == l.sum() and b.sum() == r.sum()      # smoothing conserves the totals
        for trial in range(10):
            o = dict(do_balance=bool(rs.rand() < 0.5), do_positive=bool(rs.rand() < 0.3), do_smooth=bool(rs.rand() < 0.6),
                     summary=['min', 'average', 'median'][rs.randint(3)], threshold=[0, 1, 2.5][rs.randint(3)],
                     do_scale=bool(rs.rand() < 0.5), down=bool(rs.rand() < 0.5),
                     metric=['prod', 'sum', 'euclidean', 'cosine'][rs.randint(4)])
            opt = _native.DistanceOptions(do_balance=o['do_balance'], do_positive=o['do_positive'], do_smooth=o['do_smooth'],
                                          summary=code[o['summary']], threshold=o['threshold'], do_scale=o['do_scale'],
                                          down=o['down'], metric=metric[o['metric']])
            v = ctx.profile_distance(l, r, k, opt)
            e = oracle.profile_distance(l, r, k, **o)
            assert close(v, e), (k, o, v, e)


def test_option_matrix_text_and_custom_callables(golden_scalars):
    """distance_matrix with options: device pipeline per pair == per-pair distances; a user-supplied
    summary callable takes the NumPy path and agrees with the built-in it mimics."""
    from kpal_amd import kdistlib, klib, metrics
    sets, names = golden_scalars['G8']['sets'], golden_scalars['G8']['names']
    profs = [klib.Profile.from_sequences(s, 8, name=n) for s, n in zip(sets, names)]
    d = kdistlib.ProfileDistance(do_balance=True, do_smooth=True, summary=np.median, threshold=1, do_scale=True)
    buf = io.StringIO()
    kdistlib.distance_matrix(profs, buf, 10, d)
    lines = buf.getvalue().split('\n')
    assert lines[0] == '5' and lines[1:6] == names
    at = 6
    for i in range(1, 5):
        row = [float(x) for x in lines[at].split(' ')]
        at += 1
        for j in range(i):
            e = oracle.profile_distance(profs[i].counts, profs[j].counts, 8, do_balance=True, do_smooth=True,
                                        summary='median', threshold=1, do_scale=True)
            assert abs(row[j] - e) <= 1e-9
    custom = kdistlib.ProfileDistance(do_smooth=True, summary=lambda v: np.min(v))
    assert custom._native_options() is None
    builtin = kdistlib.ProfileDistance(do_smooth=True)
    assert close(custom.distance(profs[0], profs[1]), builtin.distance(profs[0], profs[1]))


def test_option_errors():
    from kpal_amd import _native
    ctx = _native.context()
    v = np.ones(16, dtype=np.int64)
    with pytest.raises(ValueError):
        ctx.profile_distance(v, v, 2, _native.DistanceOptions(metric=7))
    with pytest.raises(ValueError):
        ctx.profile_distance(v, v, 2, _native.DistanceOptions(do_smooth=1, summary=5))
    with pytest.raises(ValueError):
        ctx.profile_distance(v, v, 3, _native.DistanceOptions())
