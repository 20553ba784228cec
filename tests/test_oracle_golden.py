"""Pins oracle/ (the CPU restatement) to golden vectors generated from the unmodified
reference (tools/gen_golden.py).  CPU only."""
import hashlib
import io
import os

import numpy as np
import pytest

import oracle
from conftest import dense


def sha(v):
    return hashlib.sha256(np.ascontiguousarray(v, dtype='<i8').tobytes()).hexdigest()


def test_g1_fixture_counts(golden_counts):
    for case in golden_counts['G1']:
        got = oracle.from_sequences(case['sequences'], case['k'])
        np.testing.assert_array_equal(got, dense(case['counts']))
        assert got.sum() == case['total'] and np.count_nonzero(got) == case['non_zero']


def test_g2_randomized_counts_and_flat_identity(golden_counts):
    assert len(golden_counts['G2']) >= 300
    for case in golden_counts['G2']:
        want = dense(case['counts'])
        np.testing.assert_array_equal(oracle.from_sequences(case['sequences'], case['k']), want)
        # joining with one separator byte gives the same windows
        flat = '\n'.join(case['sequences'])
        np.testing.assert_array_equal(oracle.count_flat(flat, case['k']), want)


def test_g3_config1_and_generator(golden_synth):
    g = golden_synth['config1']
    buf = oracle.synth_reads(g['seed'], 0, g['n_reads'], g['read_len'])
    assert buf[:150].tobytes().decode() == g['read0']
    assert buf[151 * 9999:151 * 9999 + 150].tobytes().decode() == g['read9999']
    c = oracle.count_flat(buf, g['k'])
    assert c.sum() == g['total'] == 1420000
    assert np.count_nonzero(c) == g['non_zero']
    assert list(c[:64]) == g['first64'] and list(c[-64:]) == g['last64']
    assert sha(c) == g['sha256']
    # threaded variant (private histograms + merge) is the at-scale oracle
    assert sha(oracle.count_flat(buf, g['k'], threads=4)) == g['sha256']
    # threaded variant on a table above the private-histogram limit: one shared table, atomic adds
    assert np.array_equal(oracle.count_flat(buf, 11, threads=4), oracle.count_flat(buf, 11))
    # the pure-Python restatement of klib.py:149-170 that bench.py times as the reference-speed figure
    from oracle import pyref
    reads = [bytes(r).decode() for r in buf.reshape(-1, 151)[:, :150]]
    assert sha(pyref.from_sequences(reads, g['k'])) == g['sha256']
    noisy = oracle.synth_reads(7, 0, 300, 150, noisy=True)
    seqs = [bytes(r).decode() for r in noisy.reshape(-1, 151)[:, :150]] + ['', 'ACG', 'nnnACGTacgtRYACGTT']
    for k in (1, 4, 6):
        np.testing.assert_array_equal(pyref.from_sequences(seqs, k), oracle.from_sequences(seqs, k))


def test_g3_block_counter_against_the_pinned_count(golden_synth):
    """oracle.count_blocks (the at-scale checker of tables too large for the host, tests/test_gpu_count.py::test_full_size_k15)
    against the golden-pinned count_flat / balance on BASELINE config 1 and on noisy reads: every geometry, selections that
    include the first and the last block, a block together with the block of its reverse complements, palindromes (even k)."""
    g = golden_synth['config1']
    buf = oracle.synth_reads(g['seed'], 0, g['n_reads'], g['read_len'])
    noisy = oracle.synth_reads(7, 0, 3000, 150, noisy=True)
    for data, k, block_bits, blocks, threads in (
            (buf, g['k'], 10, [0, 255, 17, 128, 77], 1),       # k = 9: 256 blocks of 1024 entries; 255 = rc block of 0
            (buf, g['k'], 10, [0, 255, 17, 128, 77], 4),
            (buf, g['k'], 18, [0], 3),                          # one block = the whole table
            (buf, g['k'], 2, list(range(0, 65536, 4099)) + [65535], 2),
            (noisy, 8, 6, [0, 1023, 512, 300, 1, 682], 3),      # even k: palindromes; 682 = 0b1010101010 (GGGGG)
            (noisy, 5, 4, [63, 0, 21], 2),
            (noisy, 11, 12, [0, 1023, 5, 1000], 4)):
        full = oracle.count_flat(data, k)
        if k == g['k'] and data is buf:
            assert sha(full) == g['sha256']
        bal = oracle.balance(full, k, threads=1)
        plain, mirror = oracle.count_blocks(data, k, block_bits, blocks, threads=threads)
        size = 1 << block_bits
        for s, b in enumerate(blocks):
            np.testing.assert_array_equal(plain[s], full[b * size:(b + 1) * size])
            np.testing.assert_array_equal(plain[s] + mirror[s], bal[b * size:(b + 1) * size])
    with pytest.raises(ValueError):
        oracle.count_blocks(buf, 9, 10, [0, 0])       # a block listed twice
    with pytest.raises(ValueError):
        oracle.count_blocks(buf, 9, 9, [0])           # not whole digits


def test_g3_noisy_and_long(golden_synth):
    g = golden_synth['noisy']
    buf = oracle.synth_reads(g['seed'], 0, g['n_reads'], 150, noisy=True)
    assert buf[:150].tobytes().decode() == g['read0']
    for case in g['cases']:
        c = oracle.count_flat(buf, case['k'])
        assert (int(c.sum()), int(np.count_nonzero(c)), sha(c)) == (case['total'], case['non_zero'], case['sha256'])
        assert sha(oracle.count_flat(buf, case['k'], threads=3)) == case['sha256']
    g = golden_synth['long']
    buf = oracle.synth_reads(g['seed'], 0, g['n_reads'], 150, noisy=True)
    seq = buf.reshape(-1, 151)[:, :150].reshape(-1)   # one long record, newlines deleted
    for case in g['cases']:
        c = oracle.from_sequences([seq], case['k'])
        assert (int(c.sum()), sha(c)) == (case['total'], case['sha256'])
        assert sha(oracle.count_flat(seq, case['k'], threads=5)) == case['sha256']


def test_g5_rc_balance_split(golden_scalars, golden_vectors):
    g = golden_scalars['G5']
    for rec in g['rc']:
        assert [oracle.reverse_complement(x, rec['k']) for x in rec['x']] == rec['rc']
    for rec in g['balance_split']:
        name, k = rec['name'], rec['k']
        v = golden_vectors['g5_%s_in' % name]
        np.testing.assert_array_equal(oracle.balance(v, k), golden_vectors['g5_%s_bal' % name])
        f, r = oracle.split(v, k)
        np.testing.assert_array_equal(f, golden_vectors['g5_%s_fwd' % name])
        np.testing.assert_array_equal(r, golden_vectors['g5_%s_rev' % name])


def _counts(seqs, k):
    return oracle.from_sequences(seqs, k)


def test_g6_known_answers(golden_scalars, golden_counts):
    g = golden_scalars['G6']
    a = _counts(g['toy_k2']['a'], 2)
    b = _counts(g['toy_k2']['b'], 2)
    assert oracle.distance(a, b, 2) == g['toy_k2']['distance'] == 0.0625
    fx = {c['fixture']: c['sequences'] for c in golden_counts['G1']}
    left = _counts(fx['LENGTH_60'], 8)
    right = _counts(fx['LENGTH_60_MORE'], 8)
    o = g['left_right_k8']
    assert oracle.distance(left, right, 8) == pytest.approx(o['prod'], rel=1e-14)
    assert o['prod'] == pytest.approx(0.4626209322, abs=1e-10)   # tests/test_kdistlib.py:114-122
    assert oracle.distance(left, right, 8, metric='sum') == pytest.approx(o['sum'], rel=1e-14)
    assert oracle.distance(left, right, 8, do_balance=True) == pytest.approx(o['balance_prod'], rel=1e-14)
    assert oracle.distance(left, right, 8, do_balance=True, metric='sum') == pytest.approx(o['balance_sum'], rel=1e-14)
    assert oracle.distance(left, right, 8, metric='euclidean') == o['euclidean']
    assert oracle.distance(left, right, 8, do_balance=True, metric='euclidean') == o['balance_euclidean']
    assert oracle.strand_balance(left, 8) == pytest.approx(o['showbalance_left'], rel=1e-14)
    assert oracle.strand_balance(left, 8, 'sum') == pytest.approx(o['showbalance_left_sum'], rel=1e-14)


def test_g7_metrics(golden_scalars, golden_vectors):
    bitexact = 0
    for rec in golden_scalars['G7']:
        l = golden_vectors['g7_%s_l' % rec['name']]
        r = golden_vectors['g7_%s_r' % rec['name']]
        for pw in ('prod', 'sum'):
            d, m = oracle.multiset(l, r, pw, return_m=True)
            assert m == rec['m']
            if np.isfinite(rec[pw]):
                assert d == pytest.approx(rec[pw], rel=1e-13, abs=1e-300)
                bitexact += d == rec[pw]
            else:
                assert not np.isfinite(d) or np.isnan(d)
        if 'euclidean' in rec:
            d, dot = oracle.euclidean(l, r, return_dot=True)
            assert dot == rec['dot']
            assert d == rec['euclidean'] or (np.isnan(d) and np.isnan(rec['euclidean']))
    assert bitexact >= 10   # the restated NumPy pairwise summation is normally bit-identical


def test_g8_matrix_text(golden_scalars):
    g = golden_scalars['G8']
    profs = [_counts(s, g['k']) for s in g['sets']]
    for case in g['cases']:
        n = case['count']
        vals = oracle.distance_matrix_values(profs[:n], g['k'], case['do_balance'], case['pairwise']) if n > 1 else []
        assert oracle.distance_matrix_text(g['names'][:n], vals, case['precision']) == case['text']


def test_g4_tutorial(golden_scalars, tutorial_dir):
    g = golden_scalars['G4']
    prof = {}
    for fname, rec in g['files'].items():
        if not fname.endswith('.fa'):
            continue
        # FASTA flattening for the shapes the reference pins: header lines dropped,
        # record lines joined, records separated (doc/tutorial.rst:44-82)
        seqs, cur = [], None
        with open(os.path.join(tutorial_dir, fname)) as fh:
            for line in fh:
                if line.startswith('>'):
                    cur = []
                    seqs.append(cur)
                elif cur is not None:
                    cur.append(line.strip())
        c = oracle.from_sequences([''.join(s) for s in seqs], 8)
        assert (int(c.sum()), int(np.count_nonzero(c)), sha(c)) == (rec['total'], rec['non_zero'], rec['sha256'])
        prof[fname[:-3]] = c
    assert g['files']['a_1.fa']['total'] == 18600 and g['files']['a_1.fa']['non_zero'] == 16141
    assert oracle.distance(prof['c_1'], prof['c_2'], 8) == pytest.approx(g['distance_c1_c2'], rel=1e-14)
    merged = [prof[s + '_1'] + prof[s + '_2'] for s in 'abcd']
    vals = oracle.distance_matrix_values(merged, 8)
    assert oracle.distance_matrix_text(list('abcd'), vals, 3) == g['matrix_abcd_p3']
    assert oracle.distance(merged[0], merged[1], 8, do_balance=True) == pytest.approx(g['distance_balanced_a_b'], rel=1e-14)
    assert oracle.strand_balance(merged[0], 8) == pytest.approx(g['showbalance_a'], rel=1e-14)


def test_g9_profile_distance_options(golden_options):
    """Oracle restatement of positive / dynamic smoothing / scaling / cosine == the reference on the
    G9 option grid (672 cases); smoothed vectors bit-exact."""
    import math
    g, z = golden_options
    for c in g['cases']:
        l, r = z['g9_%d_l' % c['pair']], z['g9_%d_r' % c['pair']]
        v = oracle.profile_distance(l, r, c['k'], c['do_balance'], c['do_positive'], c['do_smooth'], c['summary'],
                                    c['threshold'], c['do_scale'], c['down'], c['metric'])
        assert abs(v - c['distance']) <= 1e-12 * abs(c['distance']) + 1e-300, c
    checked = 0
    for pi in range(g['n_pairs']):
        for name, fn, th in g['smoothed']:
            key = 'g9_%d_%s_l' % (pi, name)
            if key not in z:
                continue
            k = int(round(math.log(z['g9_%d_l' % pi].size, 4)))
            a, b = oracle.dynamic_smooth(z['g9_%d_l' % pi], z['g9_%d_r' % pi], k, fn, th)
            np.testing.assert_array_equal(a, z[key])
            np.testing.assert_array_equal(b, z['g9_%d_%s_r' % (pi, name)])
            checked += 1
    assert checked >= 20


def test_g10_summaries_merge_shrink(golden_summaries):
    """Profile.total/non_zero/mean/median/std, merge with every built-in merger and shrink
    (kpal/klib.py:193-225,269-283,329-352): oracle vs the reference's outputs."""
    g, z = golden_summaries
    assert g['n'] == len(g['cases']) >= 15
    for i, c in enumerate(g['cases']):
        v = z['g10_%d' % i]
        s = oracle.stats(v)
        assert (s['total'], s['non_zero']) == (c['total'], c['non_zero']), i
        assert s['median'] == c['median'], i
        for key in ('mean', 'std'):
            assert abs(s[key] - c[key]) <= 1e-12 * abs(c[key]) + 1e-300, (i, key, s[key], c[key])
        assert c['shrink'] == list(range(1, c['k']))
        for factor in c['shrink']:
            np.testing.assert_array_equal(oracle.shrink(v, c['k'], factor), z['g10_%d_shrink%d' % (i, factor)])
    assert len(g['merges']) >= 20
    for m in g['merges']:
        l, r = z['g10_%d' % m['left']], z['g10_%d' % m['right_reversed']][::-1]
        np.testing.assert_array_equal(oracle.merge(l, r, m['merger']), z[m['key']], err_msg=m['key'])


@pytest.mark.parametrize('target', ['sanitize_asan', 'sanitize_tsan'])
def test_oracle_sanitized(target, tmp_path):
    """the CPU restatement under AddressSanitizer + UBSan, and its N-thread count (private
    histograms + parallel merge; shared table with relaxed atomic adds) under ThreadSanitizer (oracle/Makefile targets
    sanitize_asan / sanitize_tsan, driver oracle/sanitize_driver.c).  CPU only: sanitizers are not available on the GPU pool."""
    import shutil
    import subprocess
    odir = os.path.dirname(os.path.abspath(oracle.__file__))
    exe = str(tmp_path / target)
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    flags = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined'] if target == 'sanitize_asan' else ['-fsanitize=thread']
    build = subprocess.run(['gcc', '-O1', '-g', '-std=c11', '-fno-omit-frame-pointer'] + flags +
                           ['-o', exe, os.path.join(odir, 'sanitize_driver.c'), os.path.join(odir, 'kpal_oracle.c'), '-lm', '-lpthread'],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if build.returncode != 0 and b'sanitize' in build.stdout.lower() and (b'cannot find' in build.stdout or b'not supported' in build.stdout):
        pytest.skip('sanitizer runtime not installed: %s' % build.stdout.decode()[-300:])
    assert build.returncode == 0, build.stdout.decode()[-2000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1', TSAN_OPTIONS='halt_on_error=1 exitcode=66')
    env.pop('LD_PRELOAD', None)
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=600)
    out = run.stdout.decode()
    if run.returncode != 0 and 'unexpected memory mapping' in out:
        pytest.skip('ThreadSanitizer cannot map its shadow here (ASLR setting of the container)')
    assert run.returncode == 0 and 'SANITIZE_OK' in out, out[-3000:]
