"""The HEX pipeline (k = 12: items of six overlapping k-mers in three bytes, kpal_amd/csrc/hex_kernels.hpp) against the oracle:
every size around its geometry (48-byte lanes, 3 KiB wave-steps, tiles), every alignment of the fed buffer, halo pieces of a host
feed, reads with N / lower case (half items), homopolymers and low-complexity reads (spill list, hot-item table, drained 16-bit
bins), forced tile sizes, the staged and the atomic merge, count + balance.  (tests/test_gpu_count.py runs its strategy loops
over 'partition_hex' as well; tests/native/hex_index_check.cpp checks the index arithmetic on the CPU.)"""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

K = 12


@pytest.fixture(scope='module')
def ctx():
    from kpal_amd import _native
    return _native.context()


def count(ctx, buf, strategy='partition_hex', balance=False):
    ctx.count_begin(K, strategy)
    ctx.count_feed(buf)
    if balance:
        ctx.count_balance()
    got = ctx.count_finish()
    return got


def test_hex_sizes_and_alignments(ctx):
    base = oracle.synth_reads(31, 0, 3000, 150, noisy=True)
    for n in (0, 1, 11, 12, 13, 47, 48, 49, 95, 96, 97, 3071, 3072, 3073, 6 * 3072 + 5, 16 * 4 * 3072, 16 * 4 * 3072 + 1, 200_000, base.size):
        buf = base[:n]
        np.testing.assert_array_equal(count(ctx, buf), oracle.count_flat(buf, K), err_msg='n=%d' % n)
    # every alignment of a device buffer, with bytes around it that must not be counted
    d = ctx.alloc(70000)
    try:
        pad = np.frombuffer(b'ACGT' * 17500, dtype=np.uint8)
        for off in list(range(0, 49)) + [63, 64, 65, 191, 193]:
            n = 40_000 + off % 7
            ctx.h2d(d, pad)
            ctx.h2d(d + off, base[:n])
            ctx.count_begin(K, 'partition_hex')
            ctx.count_feed_device(d + off, n)
            assert ctx.count_last_plan()[0] == 'partition_hex'
            np.testing.assert_array_equal(ctx.count_finish(), oracle.count_flat(base[:n], K), err_msg='offset %d' % off)
    finally:
        ctx.free(d)


def test_hex_reads_and_skew(ctx):
    rs = np.random.RandomState(12)
    cases = {
        'reads_150': oracle.synth_reads(2, 0, 400_000, 150),
        'reads_noisy': oracle.synth_reads(3, 0, 300_000, 150, noisy=True),
        'reads_37': oracle.synth_reads(4, 0, 500_000, 37),
        'reads_13': oracle.synth_reads(5, 0, 500_000, 13),
        'homopolymer': np.full(30_000_000, ord('A'), dtype=np.uint8),
        'two_letter': np.frombuffer(b'AC' * 10_000_000, dtype=np.uint8),
        'period3': np.frombuffer(b'ACG' * 7_000_000, dtype=np.uint8),
        'unbroken_random': np.frombuffer(b'ACGT', dtype=np.uint8)[rs.randint(0, 4, 40_000_000)],
    }
    low = oracle.synth_reads(6, 0, 300_000, 150).reshape(-1, 151).copy()
    hit = rs.rand(low.shape[0]) < 0.05
    low[hit, :150] = ord('T')
    cases['low_complexity_5pct'] = low.reshape(-1)
    ad = oracle.synth_reads(7, 0, 300_000, 150).reshape(-1, 151).copy()
    ad[:, :34] = np.frombuffer(b'AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC', dtype=np.uint8)
    cases['adapter_prefixed'] = ad.reshape(-1)
    for name, buf in cases.items():
        want = oracle.count_flat(buf, K, threads=8, mode='private')
        np.testing.assert_array_equal(count(ctx, buf), want, err_msg=name)
        np.testing.assert_array_equal(count(ctx, buf, balance=True), oracle.balance(want, K), err_msg=name + ' balanced')
    # the whole-buffer device feed (what bench.py times), several feeds into one count
    buf = cases['reads_150']
    d = ctx.alloc(buf.size)
    try:
        ctx.h2d(d, buf)
        ctx.count_begin(K, 'partition_hex')
        ctx.count_feed_device(d, buf.size)
        ctx.count_feed_device(d, buf.size // 2)
        ctx.count_feed(cases['reads_noisy'])
        got = ctx.count_finish()
        want = oracle.count_flat(buf, K, threads=8) + oracle.count_flat(buf[:buf.size // 2], K, threads=8) + oracle.count_flat(cases['reads_noisy'], K, threads=8)
        np.testing.assert_array_equal(got, want)
    finally:
        ctx.free(d)


@pytest.mark.parametrize('steps', ['1', '2', '3', '4'])
def test_hex_forced_tiles(steps, monkeypatch):
    """Every tile size (KPAL_QUAD_STEPS) on noisy reads with a homopolymer stretch inside."""
    from kpal_amd import _native
    monkeypatch.setenv('KPAL_QUAD_STEPS', steps)
    c = _native.Context(_native.default_device())
    monkeypatch.delenv('KPAL_QUAD_STEPS')
    try:
        buf = oracle.synth_reads(40 + int(steps), 0, 250_000, 150, noisy=True).copy()
        buf[5_000_000:9_000_000] = ord('G')
        c.count_begin(K, 'partition_hex')
        c.count_feed(buf)
        assert c.count_last_plan() == ('partition_hex', int(steps), 0)
        np.testing.assert_array_equal(c.count_finish(), oracle.count_flat(buf, K, threads=8, mode='private'))
    finally:
        c.close()
