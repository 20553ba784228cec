"""Profile summaries, merge and shrink on the GPU (kpal_stats / kpal_merge / kpal_shrink; kpal/klib.py:193-225,
269-283,329-352) against the reference goldens (G10) and the oracle.  Integers (total, non_zero, median, merge,
shrink) bit-exact; mean and std within 1e-9 relative.  Run on the GPU box: pytest -m gpu."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

RTOL = 1e-9


@pytest.fixture(scope='module')
def ctx():
    from kpal_amd import _native
    return _native.context()


def check_stats(got, want, what, v=None):
    assert (got.total, got.non_zero, got.min, got.max) == (want['total'], want['non_zero'], want['min'], want['max']), what
    assert got.median == want['median'], (what, got.median, want['median'])
    # mean: 1e-9 relative to the magnitude of the data (NumPy's float64 sum of +-2^63 values cancels
    # differently from the exact 128-bit sum of the kernel; for counts, which are >= 0, this IS 1e-9 relative)
    scale = abs(want['mean']) if v is None else float(np.abs(v.astype(np.float64)).mean())
    assert abs(got.mean - want['mean']) <= RTOL * scale + 1e-300, (what, 'mean', got.mean, want['mean'])
    assert abs(got.std - want['std']) <= RTOL * abs(want['std']) + 1e-300, (what, 'std', got.std, want['std'])
    if v is not None:   # and the exact mean
        exact = sum(int(x) for x in v[:4096]) / min(v.size, 4096) if v.size <= 4096 else None
        if exact is not None:
            assert abs(got.mean - exact) <= 1e-15 * scale + 1e-300, (what, got.mean, exact)


def test_golden_summaries_through_the_profile_api(ctx, golden_summaries):
    from kpal_amd import klib, metrics
    g, z = golden_summaries
    for i, c in enumerate(g['cases']):
        v = z['g10_%d' % i]
        p = klib.Profile(v.copy())
        assert (int(p.total), int(p.non_zero)) == (c['total'], c['non_zero'])
        assert isinstance(p.median, np.float64) and p.median == c['median'], i
        assert abs(p.mean - c['mean']) <= RTOL * abs(c['mean']) + 1e-300, i
        assert abs(p.std - c['std']) <= RTOL * abs(c['std']) + 1e-300, i
        np.testing.assert_array_equal(p.counts, v)                      # summaries do not modify the counts
        for factor in c['shrink']:
            q = klib.Profile(v.copy())
            q.shrink(factor)
            assert q.length == c['k'] - factor and q.counts.dtype == np.int64
            np.testing.assert_array_equal(q.counts, z['g10_%d_shrink%d' % (i, factor)])
    for m in g['merges']:
        p = klib.Profile(z['g10_%d' % m['left']].copy())
        q = klib.Profile(z['g10_%d' % m['right_reversed']][::-1].copy())
        before = q.counts.copy()
        p.merge(q, merger=metrics.mergers[m['merger']])
        np.testing.assert_array_equal(p.counts, z[m['key']], err_msg=m['key'])
        np.testing.assert_array_equal(q.counts, before)
    p = klib.Profile(np.arange(16, dtype=np.int64))
    q = p.copy()
    q.counts[0] = 99
    p.merge(q)                                                           # default merger: sum
    assert p.counts[0] == 99 and p.counts[1] == 2
    p.shrink()
    assert p.length == 1 and list(p.counts) == [111, 44, 76, 108]
    with pytest.raises(ValueError):
        p.shrink(1)
    p.shrink(0)
    assert p.length == 1 and list(p.counts) == [111, 44, 76, 108]


def test_stats_random_vectors_vs_oracle(ctx):
    rs = np.random.RandomState(31)
    cases = []
    for k in (1, 2, 5, 8, 10):
        n = 4 ** k
        cases.append(('poisson k=%d' % k, rs.poisson(rs.choice([0.02, 0.7, 3.0, 800.0]), n).astype(np.int64)))
    cases.append(('zeros', np.zeros(4 ** 6, dtype=np.int64)))
    cases.append(('one nonzero', np.eye(1, 4 ** 6, 77, dtype=np.int64)[0] * 5))
    cases.append(('negative', rs.randint(-10 ** 6, 10 ** 6, size=4 ** 7).astype(np.int64)))
    cases.append(('small negative sum', np.array([-3, 1, 0, 1], dtype=np.int64)))
    cases.append(('wide', rs.randint(-(1 << 62), 1 << 62, size=4 ** 8).astype(np.int64)))
    cases.append(('wrapping total', rs.randint(1 << 60, 1 << 62, size=4 ** 6).astype(np.int64)))
    cases.append(('extremes', np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max, 0, -1] * 4, dtype=np.int64)))
    cases.append(('two values', np.repeat(np.array([3, 1 << 40], dtype=np.int64), 512)))
    cases.append(('odd length', rs.poisson(2.0, 1001).astype(np.int64)))       # the C-ABI takes any n >= 1
    cases.append(('single', np.array([42], dtype=np.int64)))
    heavy = rs.poisson(1.5, 4 ** 9).astype(np.int64)
    heavy[rs.randint(0, heavy.size, 1000)] = 10 ** 12                          # long tail: every select byte is used
    cases.append(('heavy tail', heavy))
    for what, v in cases:
        check_stats(ctx.stats(v), oracle.stats(v), what, v)
        # and NumPy, the arithmetic the reference calls (the oracle restates it)
        assert ctx.stats(v).median == float(np.median(v)), what
    with pytest.raises(ValueError):
        ctx.stats(np.zeros(0, dtype=np.int64))


def test_stats_of_a_counted_table_on_the_device(ctx):
    """kpal_stats_device on the table of a count, without the 128 MiB download (k = 12)."""
    n_reads = 200000
    buf = oracle.synth_reads(5, 0, n_reads, 150, noisy=True)
    ctx.count_begin(12)
    ctx.count_feed(buf)
    dev, bins = ctx.count_table()
    got = ctx.stats_device(dev, bins)
    table = ctx.count_finish()
    want = oracle.stats(table)
    check_stats(got, want, 'k=12 table')
    assert got.total == int(oracle.count_flat(buf, 12, threads=8).sum())


def test_merge_and_shrink_random_vs_oracle(ctx):
    rs = np.random.RandomState(37)
    for k in (1, 3, 6, 9, 11):
        n = 4 ** k
        l = rs.poisson(0.8, n).astype(np.int64)
        r = rs.poisson(0.8, n).astype(np.int64)
        if k == 6:
            l = rs.randint(-(1 << 62), 1 << 62, size=n).astype(np.int64)      # sums wrap
            r = rs.randint(-(1 << 62), 1 << 62, size=n).astype(np.int64)
            r[::3] = 0
        for code, name in enumerate(('sum', 'xor', 'int', 'nint')):
            np.testing.assert_array_equal(ctx.merge(l, r, code), oracle.merge(l, r, name), err_msg='%s k=%d' % (name, k))
        for factor in range(1, k):
            np.testing.assert_array_equal(ctx.shrink(l, k, factor), oracle.shrink(l, k, factor), err_msg='k=%d f=%d' % (k, factor))
    with pytest.raises(ValueError):
        ctx.merge(np.zeros(4, dtype=np.int64), np.zeros(16, dtype=np.int64), 0)
    with pytest.raises(ValueError):
        ctx.shrink(np.zeros(16, dtype=np.int64), 2, 2)
    with pytest.raises(ValueError):
        ctx.merge(np.zeros(4, dtype=np.int64), np.zeros(4, dtype=np.int64), 7)
