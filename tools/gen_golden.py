#!/opt/conda/bin/python3.9
"""Generate tests/golden/* by importing the UNMODIFIED reference (kPAL at /root/reference).

Run in the build container only (the reference never travels to the GPU box):

    PYTHONPATH=tools/oracle_stubs:/root/reference /opt/conda/bin/python3.9 tools/gen_golden.py

tools/oracle_stubs holds import shims for two third-party packages the image lacks
(Bio.SeqIO, semantic_version); everything that computes an expected value below is the
reference's own code: kpal.klib.Profile, kpal.metrics, kpal.kdistlib, kpal.kmer.

Fixture groups follow SURVEY.md section 8c (G1..G8); G9 pins every ProfileDistance option
(balance, positive, dynamic smoothing with each summary function, scaling, every metric:
kpal/kdistlib.py:126-161); G10 pins the profile summaries, Profile.merge with every built-in
merger and Profile.shrink (kpal/klib.py:193-225,269-283,329-352); G11 pins the callers of section 8
row a14 (kmer.count/merge/balance/get_balance/get_stats/distance/distance_matrix) through real HDF5
files; G12 pins the command line: kpal.kmer.main([...]) for every sub-command on the tutorial files
(kpal/kmer.py:703-975) -- stdout / text outputs, the stored counts (sha256), dataset and file attributes,
and the usage errors.  Only DATA is written: inputs and the reference's outputs.

    ... tools/gen_golden.py            # everything
    ... tools/gen_golden.py g12        # only tests/golden/cli.json
"""
from __future__ import print_function

import hashlib
import io
import json
import os
import random
import sys
import zipfile

import numpy as np

from kpal import kdistlib, klib, kmer, metrics

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'tests', 'golden')

# Fixture sequences of the reference's own tests (tests/utils.py:24-59) -- data.
LENGTH_8 = ['GTACATGA', 'TAAACTAA', 'TATCTTTA', 'TACTATGT']
LENGTH_8_WITH_N = ['GNACATGA', 'TAAACTNA', 'TATNTTTA', 'TACTATGN']
LENGTH_60 = ['GTACATGATAGGTCCACAGCTCTGAGCAAGGCAGACGTCCATACTTAAAACCCAGACTGC',
             'TAAACTAAAAGAAAGAATTTTTTTAATGGTAGACTACCTAAAATTATGTCTCTTAGTCCT',
             'TATCTTTACCTATATATTTGACTAAGATTTAGTATTACTACTACCTAAAATTATGTCTCT',
             'TACTATGTCTTGAAGGACAGCACCTGACCTCCCCCTGCAAGGTGTCATCCCCAAGCTGGT']
LENGTH_60_WITH_N = ['GTANATGATAGGTCCACAGCTCTGAGCAAGGCAGACGTCCATACTTAAAACCCAGACTGC',
                    'TAAACTAAAAGAAAGAATTTTTTTAATGGTAGACTACCTAAAATTATGTCTCTTAGNCCT',
                    'TATCTTTACCTATATATTTGACTAAGATTTAGTNTTACTACTACCTAAAATTATGTCTCT',
                    'TACTATGTCTTGAAGGACAGCCCTGACCTCCCCCTGCAAGGTGTCATCCCCAAGCTGGTN']
LENGTH_60_MORE = ['TTACAATGATTAGGTCCACAGCTCTGAGCAACGCGCAGACGTCACATACTTCAAAACCCA',
                  'TAAAAACTATATAAGAAATCGAATTTTCTCTTAATGGTAGCAGCTACCGTAAAATCTATG',
                  'TACGCCTATATATCTTTGACTAAGCATTTATGTATTACATACTAACCAAAATTACTGTCT',
                  'TACTAAGTTTTGAAGGACAGCACATCACCTGCGCAATATCGGTGTCACCCCATAGCTCCT']
FIXTURES = {'LENGTH_8': LENGTH_8, 'LENGTH_8_WITH_N': LENGTH_8_WITH_N, 'LENGTH_60': LENGTH_60,
            'LENGTH_60_WITH_N': LENGTH_60_WITH_N, 'LENGTH_60_MORE': LENGTH_60_MORE}


def sparse(v):
    v = np.asarray(v)
    idx = np.nonzero(v)[0]
    return {'n': int(v.size), 'idx': [int(i) for i in idx], 'val': [int(x) for x in v[idx]]}


def fasta_text(seqs, width=None):
    out = []
    for i, s in enumerate(seqs):
        out.append('>r%d' % (i + 1))
        if width:
            out.extend(s[j:j + width] for j in range(0, len(s), width))
        else:
            out.append(s)
    return '\n'.join(out) + '\n'


# ---- SURVEY 8d synthetic generator, pure-Python ints (third independent statement) ----
M64 = (1 << 64) - 1


def mix64(x):
    z = (x + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def synth_read(seed, r, read_len=150, noisy=False):
    out = []
    for pos in range(read_len):
        g = r * read_len + pos
        w = mix64((seed * 0xD1342543DE82EF95 + (g >> 5)) & M64)
        c = 'ACGT'[(w >> (2 * (g & 31))) & 3]
        if noisy:
            h = mix64(((~seed) + g) & M64)
            if h % 1000 == 0:
                c = 'N'
            elif h % 100 == 1:
                c = c.lower()
        out.append(c)
    return ''.join(out)


def g1():
    cases = []
    for name, seqs in sorted(FIXTURES.items()):
        for k in (1, 4, 7, 8):
            p = klib.Profile.from_sequences(seqs, k)
            pf = klib.Profile.from_fasta(io.StringIO(fasta_text(seqs)), k)
            assert (p.counts == pf.counts).all()
            cases.append({'fixture': name, 'sequences': seqs, 'k': k, 'counts': sparse(p.counts),
                          'total': int(p.total), 'non_zero': int(p.non_zero)})
    # k == len and k == len-1 (tests/test_klib.py:72-81)
    for k in (60, 59):
        pass  # 4**60 bins is not representable; the reference tests use LENGTH_8 with k=8/7 (above)
    return cases


def g2():
    rnd = random.Random(20261002)
    alphabet = 'ACGT' * 4 + 'acgt' + 'NnRY- '
    cases = []
    for _ in range(320):
        k = rnd.randint(1, 8)
        nseq = rnd.randint(0, 6)
        seqs = [''.join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 40))) for _ in range(nseq)]
        p = klib.Profile.from_sequences(seqs, k)
        cases.append({'k': k, 'sequences': seqs, 'counts': sparse(p.counts)})
    # Appendix B quirks
    for seqs, k in ((['RYACGU'], 2), ([], 3), ([''], 3), (['AC', 'G'], 3), (['ACGT' * 10], 4),
                    (['A' * 50], 5), (['ACGTUacgtu'], 2)):
        p = klib.Profile.from_sequences(seqs, k)
        cases.append({'k': k, 'sequences': seqs, 'counts': sparse(p.counts)})
    return cases


def g3():
    out = {}
    reads = [synth_read(1, r) for r in range(10000)]
    p = klib.Profile.from_sequences(reads, 9)
    c = p.counts.astype('<i8')
    out['config1'] = {'seed': 1, 'n_reads': 10000, 'read_len': 150, 'k': 9,
                      'sha256': hashlib.sha256(c.tobytes()).hexdigest(),
                      'total': int(p.total), 'non_zero': int(p.non_zero),
                      'first64': [int(x) for x in c[:64]], 'last64': [int(x) for x in c[-64:]],
                      'read0': reads[0], 'read9999': reads[9999]}
    # robustness variant (N + lower-case), smaller, several k
    reads = [synth_read(7, r, noisy=True) for r in range(1500)]
    out['noisy'] = {'seed': 7, 'n_reads': 1500, 'read_len': 150, 'read0': reads[0], 'cases': []}
    for k in (3, 6, 9, 11, 12):
        p = klib.Profile.from_sequences(reads, k)
        c = p.counts.astype('<i8')
        out['noisy']['cases'].append({'k': k, 'sha256': hashlib.sha256(c.tobytes()).hexdigest(),
                                      'total': int(p.total), 'non_zero': int(p.non_zero)})
    # one long mixed record (chunk seams / halo): a single 200 kb sequence
    long_seq = ''.join(synth_read(11, r, noisy=True) for r in range(1400))
    out['long'] = {'seed': 11, 'n_reads': 1400, 'cases': []}
    for k in (5, 10, 12):
        p = klib.Profile.from_sequences([long_seq], k)
        c = p.counts.astype('<i8')
        out['long']['cases'].append({'k': k, 'sha256': hashlib.sha256(c.tobytes()).hexdigest(),
                                     'total': int(p.total), 'non_zero': int(p.non_zero)})
    return out


def g4():
    """Tutorial data (doc/downloads/tutorial.zip; doc/tutorial.rst:44-144)."""
    zpath = '/root/reference/doc/downloads/tutorial.zip'
    tdir = os.path.join(OUT, 'tutorial')
    os.makedirs(tdir, exist_ok=True)
    z = zipfile.ZipFile(zpath)
    profiles = {}
    out = {'k': 8, 'files': {}}
    for info in sorted(z.infolist(), key=lambda i: i.filename):
        if not info.filename.endswith('.fa'):
            continue
        base = os.path.basename(info.filename)
        data = z.read(info).decode('ascii')
        with open(os.path.join(tdir, base), 'w') as fh:
            fh.write(data)
        p = klib.Profile.from_fasta(io.StringIO(data), 8, name=base[:-3])
        profiles[base[:-3]] = p
        c = p.counts.astype('<i8')
        out['files'][base] = {'total': int(p.total), 'non_zero': int(p.non_zero),
                              'sha256': hashlib.sha256(c.tobytes()).hexdigest()}
    # merged a = a_1 + a_2 etc. (doc/tutorial.rst:84-102), distances (104-144)
    merged = {}
    for s in 'abcd':
        m = profiles[s + '_1'].copy()
        m.merge(profiles[s + '_2'])
        m.name = s
        merged[s] = m
        out['files'][s + '_merged'] = {'total': int(m.total), 'non_zero': int(m.non_zero)}
    d = kdistlib.ProfileDistance()
    out['distance_c1_c2'] = d.distance(profiles['c_1'], profiles['c_2'])
    buf = io.StringIO()
    kdistlib.distance_matrix([merged[s] for s in 'abcd'], buf, 3, d)
    out['matrix_abcd_p3'] = buf.getvalue()
    db = kdistlib.ProfileDistance(do_balance=True)
    out['distance_balanced_a_b'] = db.distance(merged['a'], merged['b'])
    f, r = merged['a'].split()
    out['showbalance_a'] = metrics.multiset(f, r, metrics.pairwise['prod'])
    return out


def g5(arrays):
    out = {'rc': [], 'balance_split': []}
    for k in (1, 2, 3, 4, 5, 8, 12, 15):
        p = klib.Profile(np.zeros(4 ** min(k, 8), dtype='int64'))
        p.length = k
        rnd = random.Random(k)
        xs = [0, 1, 4 ** k - 1] + [rnd.randrange(4 ** k) for _ in range(20)]
        out['rc'].append({'k': k, 'x': xs, 'rc': [int(p.reverse_complement(x)) for x in xs]})
    rs = np.random.RandomState(5)
    cases = [('AATT', ['AATT'], 4), ('ACCTAGGT', LENGTH_60 + ['ACCTAGGT'], 8), ('SEQ2', LENGTH_60, 2),
             ('SEQ8', LENGTH_60, 8)]
    for name, seqs, k in cases:
        p = klib.Profile.from_sequences(seqs, k)
        arrays['g5_%s_in' % name] = p.counts.copy()
        b = p.copy()
        b.balance()
        f, r = p.split()
        arrays['g5_%s_bal' % name] = b.counts
        arrays['g5_%s_fwd' % name] = np.asarray(f, dtype='int64')
        arrays['g5_%s_rev' % name] = np.asarray(r, dtype='int64')
        out['balance_split'].append({'name': name, 'k': k})
    for k in range(1, 7):
        v = rs.randint(0, 1 << 62, size=4 ** k).astype('int64')
        v[rs.rand(4 ** k) < 0.3] = 0
        name = 'rand%d' % k
        p = klib.Profile(v.copy())
        b = p.copy()
        b.balance()
        f, r = p.split()
        arrays['g5_%s_in' % name] = v
        arrays['g5_%s_bal' % name] = b.counts
        arrays['g5_%s_fwd' % name] = np.asarray(f, dtype='int64')
        arrays['g5_%s_rev' % name] = np.asarray(r, dtype='int64')
        out['balance_split'].append({'name': name, 'k': k})
    return out


def as_array(seqs, k):
    return klib.Profile.from_sequences(seqs, k).counts


def g6():
    out = {}
    a = ['AC', 'AG', 'AT', 'CA', 'CC', 'CG', 'CT', 'GA', 'GC', 'GG', 'GT', 'TA', 'TG', 'TT']
    b = ['AC', 'AT', 'CA', 'CC', 'CG', 'CT', 'GA', 'GC', 'GG', 'GT', 'TA', 'TC', 'TG', 'TT']
    pa = klib.Profile(as_array(a, 2))
    pb = klib.Profile(as_array(b, 2))
    out['toy_k2'] = {'a': a, 'b': b, 'distance': kdistlib.ProfileDistance().distance(pa, pb)}
    left = klib.Profile(as_array(LENGTH_60, 8), 'left')
    right = klib.Profile(as_array(LENGTH_60_MORE, 8), 'right')
    P = metrics.pairwise
    o = {}
    o['prod'] = kdistlib.ProfileDistance().distance(left, right)
    o['sum'] = kdistlib.ProfileDistance(pairwise=P['sum']).distance(left, right)
    o['balance_prod'] = kdistlib.ProfileDistance(do_balance=True).distance(left, right)
    o['balance_sum'] = kdistlib.ProfileDistance(do_balance=True, pairwise=P['sum']).distance(left, right)
    o['euclidean'] = kdistlib.ProfileDistance(distance_function=metrics.euclidean).distance(left, right)
    o['balance_euclidean'] = kdistlib.ProfileDistance(
        do_balance=True, distance_function=metrics.euclidean).distance(left, right)
    o['cosine'] = kdistlib.ProfileDistance(distance_function=metrics.cosine_similarity).distance(left, right)
    f, r = left.split()
    o['showbalance_left'] = metrics.multiset(f, r, P['prod'])
    o['showbalance_left_sum'] = metrics.multiset(f, r, P['sum'])
    o['smooth_min'] = kdistlib.ProfileDistance(do_smooth=True).distance(left, right)
    o['smooth_avg'] = kdistlib.ProfileDistance(do_smooth=True, summary=np.mean).distance(left, right)
    o['positive'] = kdistlib.ProfileDistance(do_positive=True).distance(left, right)
    o['scale'] = kdistlib.ProfileDistance(do_scale=True).distance(left, right)
    o['scale_down'] = kdistlib.ProfileDistance(do_scale=True, down=True).distance(left, right)
    out['left_right_k8'] = {k: float(v) for k, v in o.items()}
    return out


def g7(arrays):
    rs = np.random.RandomState(7)
    out = []

    def add(name, l, r):
        arrays['g7_%s_l' % name] = l
        arrays['g7_%s_r' % name] = r
        rec = {'name': name, 'n': int(l.size), 'dtype': str(l.dtype)}
        with np.errstate(all='ignore'):
            rec['prod'] = float(metrics.multiset(l, r, metrics.pairwise['prod']))
            rec['sum'] = float(metrics.multiset(l, r, metrics.pairwise['sum']))
            rec['m'] = int(np.count_nonzero(np.logical_or(l, r)))
            if l.dtype.kind == 'i':
                rec['euclidean'] = float(metrics.euclidean(l, r))
                d = np.subtract(l, r)
                rec['dot'] = int(np.dot(d, d))
        out.append(rec)

    for k in (6, 7, 8):
        n = 4 ** k
        add('dense%d' % k, rs.poisson(830, n).astype('int64'), rs.poisson(830, n).astype('int64'))
        add('sparse%d' % k, rs.poisson(0.8, n).astype('int64'), rs.poisson(0.8, n).astype('int64'))
    n = 4 ** 6
    l = rs.poisson(5, n).astype('int64')
    r = rs.poisson(5, n).astype('int64')
    l[rs.rand(n) < 0.5] = 0
    add('onesided', l, r)
    add('allzero', np.zeros(64, 'int64'), np.zeros(64, 'int64'))
    add('leftzero', np.zeros(256, 'int64'), rs.poisson(3, 256).astype('int64'))
    # values where (x+1)*(y+1) wraps int64 (metrics.py:160 evaluates in int64)
    big = rs.randint(3 * 10 ** 9, 4 * 10 ** 9, size=256).astype('int64')
    big2 = rs.randint(3 * 10 ** 9, 4 * 10 ** 9, size=256).astype('int64')
    add('wrap', big, big2)
    huge = rs.randint(1 << 40, 1 << 61, size=1024).astype('int64')
    huge2 = rs.randint(1 << 40, 1 << 61, size=1024).astype('int64')
    add('huge', huge, huge2)
    # float64 (scaled) inputs, kdistlib.py:149-157
    lf = rs.poisson(20, 1024).astype('int64') * 1.37
    rf = rs.poisson(20, 1024).astype('float64')
    lf[rs.rand(1024) < 0.2] = 0.0
    add('float', lf, rf)
    return out


def g8():
    out = []
    sets = [LENGTH_60, LENGTH_60_MORE, LENGTH_60, LENGTH_60_WITH_N, LENGTH_8 * 3]
    names = ['a', 'b', 'c', 'd', 'e']
    profs = [klib.Profile(as_array(s, 8), n) for s, n in zip(sets, names)]
    for count in (1, 2, 3, 5):
        for precision in (2, 3, 10):
            for do_balance in (False, True):
                for pw in ('prod', 'sum'):
                    d = kdistlib.ProfileDistance(do_balance=do_balance, pairwise=metrics.pairwise[pw])
                    buf = io.StringIO()
                    kdistlib.distance_matrix(profs[:count], buf, precision, d)
                    out.append({'count': count, 'precision': precision, 'do_balance': do_balance,
                                'pairwise': pw, 'text': buf.getvalue()})
    d = kdistlib.ProfileDistance(distance_function=metrics.euclidean)
    buf = io.StringIO()
    kdistlib.distance_matrix(profs, buf, 6, d)
    out.append({'count': 5, 'precision': 6, 'do_balance': False, 'pairwise': 'euclidean',
                'text': buf.getvalue()})
    return {'sets': sets, 'names': names, 'k': 8, 'cases': out}


def g9(arrays):
    """ProfileDistance option grid on random profile pairs (kdistlib.py:126-161)."""
    rs = np.random.RandomState(9)
    summaries = {'min': metrics.summary['min'], 'average': metrics.summary['average'],
                 'median': metrics.summary['median']}
    dist_funcs = {'prod': None, 'sum': None, 'euclidean': metrics.vector_distance['euclidean'],
                  'cosine': metrics.vector_distance['cosine']}
    pairs = []
    for k, lam in ((2, 3.0), (3, 0.7), (3, 6.0), (4, 0.5), (4, 2.0), (4, 40.0), (5, 0.3), (5, 1.5), (5, 12.0),
                   (6, 0.8), (6, 3.0), (7, 1.1)):
        n = 4 ** k
        l = rs.poisson(lam, n).astype('int64')
        r = rs.poisson(lam * rs.uniform(0.5, 2.0), n).astype('int64')
        if k >= 4:   # long empty stretches so that whole sub-profiles collapse
            a, b = sorted(rs.randint(0, n, 2))
            l[a:b] //= 3
            r[a // 2:b // 2] = 0
        pairs.append((k, l, r))
    cases = []
    for pi, (k, l, r) in enumerate(pairs):
        arrays['g9_%d_l' % pi] = l
        arrays['g9_%d_r' % pi] = r
        left, right = klib.Profile(l.copy(), 'l'), klib.Profile(r.copy(), 'r')
        for ci in range(56):
            o = {'pair': pi, 'k': k,
                 'do_balance': bool(rs.rand() < 0.4), 'do_positive': bool(rs.rand() < 0.3),
                 'do_smooth': bool(rs.rand() < 0.6),
                 'summary': ['min', 'average', 'median'][rs.randint(3)],
                 'threshold': [0, 1, 2, 5, 0.5, 2.5, 7.25][rs.randint(7)],
                 'do_scale': bool(rs.rand() < 0.4), 'down': bool(rs.rand() < 0.5),
                 'metric': ['prod', 'sum', 'euclidean', 'cosine'][rs.randint(4)]}
            d = kdistlib.ProfileDistance(
                do_balance=o['do_balance'], do_positive=o['do_positive'], do_smooth=o['do_smooth'],
                summary=summaries[o['summary']], threshold=o['threshold'], do_scale=o['do_scale'],
                down=o['down'], distance_function=dist_funcs[o['metric']],
                pairwise=metrics.pairwise[o['metric'] if o['metric'] in ('prod', 'sum') else 'prod'])
            with np.errstate(all='ignore'):
                o['distance'] = float(d.distance(left, right))
            assert (left.counts == l).all() and (right.counts == r).all()
            cases.append(o)
        # the smoothed vectors themselves, for three settings per pair
        if k <= 5:
            for name, fn, th in (('min0', 'min', 0), ('avg2', 'average', 2), ('med25', 'median', 2.5)):
                a, b = left.copy(), right.copy()
                kdistlib.ProfileDistance(do_smooth=True, summary=summaries[fn], threshold=th).dynamic_smooth(a, b)
                arrays['g9_%d_%s_l' % (pi, name)] = a.counts
                arrays['g9_%d_%s_r' % (pi, name)] = b.counts
    return {'n_pairs': len(pairs), 'smoothed': [['min0', 'min', 0], ['avg2', 'average', 2], ['med25', 'median', 2.5]],
            'cases': cases}


def g10(arrays):
    """Profile.total/non_zero/mean/median/std, merge with each metrics.mergers entry, shrink."""
    rs = np.random.RandomState(10)
    vecs = []
    for k, lam in ((1, 2.0), (2, 0.5), (3, 4.0), (4, 0.3), (4, 30.0), (5, 1.0), (6, 0.05), (6, 900.0), (7, 2.5)):
        vecs.append((k, rs.poisson(lam, 4 ** k).astype('int64')))
    v = rs.poisson(3.0, 4 ** 5).astype('int64')
    v[rs.rand(v.size) < 0.5] = 0                      # median 0 / 0.5 territory
    vecs.append((5, v))
    vecs.append((4, np.full(4 ** 4, 7, dtype='int64')))                                   # constant
    vecs.append((5, rs.randint(1 << 40, 1 << 61, size=4 ** 5).astype('int64')))           # sum wraps int64
    vecs.append((4, rs.randint(-1000, 1000, size=4 ** 4).astype('int64')))                # negative entries
    vecs.append((6, (rs.randint(0, 3, size=4 ** 6) * (1 << 33)).astype('int64')))         # few distinct, > 2^32
    v = np.zeros(4 ** 3, dtype='int64')
    v[:31] = 5
    v[31] = 9
    v[32:] = 11                                         # the two middle elements differ: 9 and 11
    vecs.append((3, v))
    cases = []
    for i, (k, c) in enumerate(vecs):
        arrays['g10_%d' % i] = c
        p = klib.Profile(c.copy())
        with np.errstate(all='ignore'):
            case = {'k': k, 'total': int(p.total), 'non_zero': int(p.non_zero), 'mean': float(p.mean),
                    'median': float(p.median), 'std': float(p.std), 'shrink': []}
        for factor in range(1, k):
            q = klib.Profile(c.copy())
            with np.errstate(all='ignore'), __import__('warnings').catch_warnings():
                __import__('warnings').simplefilter('ignore')
                q.shrink(factor)
            assert q.length == k - factor
            arrays['g10_%d_shrink%d' % (i, factor)] = q.counts
            case['shrink'].append(factor)
        cases.append(case)
    merges = []
    pairs = [(0, 0), (2, 2), (3, 4), (5, 9), (6, 7), (11, 9), (12, 3)]
    for a, b in pairs:
        if vecs[a][1].size != vecs[b][1].size:
            continue
        for name in ('sum', 'xor', 'int', 'nint'):
            p = klib.Profile(vecs[a][1].copy())
            q = klib.Profile(vecs[b][1][::-1].copy())
            with np.errstate(all='ignore'):
                p.merge(q, merger=metrics.mergers[name])
            key = 'g10_merge_%d_%d_%s' % (a, b, name)
            arrays[key] = np.asarray(p.counts, dtype='int64')
            merges.append({'left': a, 'right_reversed': b, 'merger': name, 'key': key})
    return {'n': len(vecs), 'cases': cases, 'merges': merges}


def g11():
    """The callers of SURVEY.md 8 row a14 -- kmer.count / merge / balance / get_balance / get_stats /
    distance / distance_matrix (kpal/kmer.py:112-271,541-700) -- on the tutorial FASTA files through
    real HDF5 files (h5py): dataset attributes, sha256 of the stored counts, and the text outputs."""
    import tempfile
    import h5py
    tdir = os.path.join(OUT, 'tutorial')
    tmp = tempfile.mkdtemp()

    def h5(name):
        f = h5py.File(os.path.join(tmp, name), 'w')
        f.create_group('profiles')
        return f

    def describe(f):
        out = {}
        for name in sorted(f['profiles']):
            ds = f['profiles/' + name]
            out[name] = {'attrs': dict((k, float(v) if isinstance(v, (float, np.floating)) else int(v))
                                       for k, v in ds.attrs.items()),
                         'sha256': hashlib.sha256(ds[:].astype('<i8').tobytes()).hexdigest()}
        return out

    def text(fn, *args, **kw):
        buf = io.StringIO()
        fn(*args, **kw)
        return buf

    out = {}
    files = ['a_1', 'a_2', 'b_1', 'b_2', 'c_1', 'c_2']
    handles = [open(os.path.join(tdir, n + '.fa')) for n in files]
    counted = h5('count.k8')
    kmer.count(handles, counted, 8)                         # names from the file names
    out['count_k8'] = describe(counted)
    for h in handles:
        h.seek(0)
    named = h5('named.k5')
    kmer.count(handles[:2], named, 5, names=['x', 'y'])
    out['count_k5_named'] = describe(named)
    for h in handles:
        h.seek(0)
    rec1 = h5('rec1.k4')
    one = '>first some title\nACGTTGCAACGT\nACG\n>second\nNNACGTN\n>third\nAC\n'
    kmer.count([io.StringIO(one)], rec1, 4, by_record=True)  # one file: no prefix
    out['by_record_one_file'] = {'input': one, 'profiles': describe(rec1)}
    rec2 = h5('rec2.k4')
    small = [io.StringIO('>r1 x\nACGTACGTAA\n>r2\nTTTTT\nGGGNAC\n'), io.StringIO('>r1\nCCCCCCC\n')]
    kmer.count(small, rec2, 3, names=['p', 'q'], by_record=True)
    out['by_record_two_files'] = {'inputs': ['>r1 x\nACGTACGTAA\n>r2\nTTTTT\nGGGNAC\n', '>r1\nCCCCCCC\n'],
                                  'profiles': describe(rec2)}
    # merge: left a_1, b_1 with right a_2, b_2 (names differ -> concatenated), every built-in merger
    out['merge'] = {}
    for merger in ('sum', 'xor', 'int', 'nint'):
        m = h5('merge_%s.k8' % merger)
        kmer.merge(counted, counted, m, names_left=['a_1', 'b_1'], names_right=['a_2', 'b_2'], merger=merger)
        out['merge'][merger] = describe(m)
    m = h5('merge_same.k8')
    kmer.merge(counted, counted, m, names_left=['c_1'], names_right=['c_1'])
    out['merge_same_name'] = describe(m)
    m = h5('merge_custom.k8')
    kmer.merge(counted, counted, m, names_left=['c_1'], names_right=['c_2'], custom_merger='np.maximum(left, right)')
    out['merge_custom'] = describe(m)
    bal = h5('balanced.k8')
    kmer.balance(counted, bal, names=['a_1', 'c_2'])
    out['balance'] = describe(bal)
    for key, fn, kw in (('get_balance_p10', kmer.get_balance, {}), ('get_balance_p3', kmer.get_balance, {'precision': 3}),
                        ('get_stats_p10', kmer.get_stats, {}), ('get_stats_p4', kmer.get_stats, {'precision': 4, 'names': ['b_2', 'a_1']})):
        buf = io.StringIO()
        fn(counted, buf, **kw)
        out[key] = buf.getvalue()
    left, right = h5('left.k8'), h5('right.k8')
    for n in ('a', 'b', 'c'):
        klib.Profile.from_file(counted, n + '_1').save(left, name=n)
        klib.Profile.from_file(counted, n + '_2').save(right, name=n)
    out['distance'] = []
    for kw in ({}, {'precision': 3}, {'do_balance': True, 'precision': 8}, {'pairwise': 'sum', 'precision': 8},
               {'distance_function': 'euclidean', 'precision': 6}, {'distance_function': 'cosine', 'precision': 8},
               {'do_smooth': True, 'summary': 'average', 'threshold': 2, 'precision': 8},
               {'do_smooth': True, 'summary': 'median', 'threshold': 1, 'do_scale': True, 'down': True, 'precision': 8},
               {'do_positive': True, 'do_scale': True, 'precision': 8},
               {'custom_pairwise': 'abs(left - right) / (left + right + 2)', 'precision': 8},
               {'do_smooth': True, 'custom_summary': 'np.max(values)', 'threshold': 3, 'precision': 8},
               {'names_left': ['c', 'a'], 'names_right': ['b', 'b'], 'precision': 8}):
        buf = io.StringIO()
        kmer.distance(left, right, buf, **kw)
        out['distance'].append({'kwargs': kw, 'text': buf.getvalue()})
    out['matrix'] = []
    for kw in ({'precision': 3}, {'precision': 8, 'do_balance': True}, {'precision': 8, 'pairwise': 'sum', 'names': ['c_2', 'a_1', 'b_1']},
               {'precision': 6, 'distance_function': 'euclidean'},
               {'precision': 8, 'do_smooth': True, 'summary': 'min', 'threshold': 1, 'do_scale': True}):
        buf = io.StringIO()
        kmer.distance_matrix(counted, buf, **kw)
        out['matrix'].append({'kwargs': kw, 'text': buf.getvalue()})
    out['files'] = files
    return out


def g12():
    """The command line (kpal/kmer.py:703-975): every sub-command through kmer.main in a scratch directory
    holding the tutorial FASTA files.  Per step: argv, exit status, stdout, the text files and the profile
    files it created (root attributes, per-profile attributes, sha256 of the stored counts; for `shuffle`
    the sha256 of the SORTED counts), and for failing steps the message after "error: "."""
    import contextlib
    import shutil
    import tempfile
    import h5py
    tdir = os.path.join(OUT, 'tutorial')
    tmp = tempfile.mkdtemp()
    for n in ('a_1', 'a_2', 'b_1', 'b_2', 'c_1', 'c_2'):
        shutil.copy(os.path.join(tdir, n + '.fa'), tmp)
    inputs = {
        'rec.fa': '>first some title\nACGTTGCAACGT\nACG\n>second\nNNACGTN\n>third\nAC\n',
        # old plaintext format: three header lines (length, total, non-zero), then one count per line
        'old1.txt': '2\n20\n9\n' + '\n'.join(str(v) for v in [3, 0, 1, 2, 0, 0, 4, 1, 0, 5, 0, 0, 2, 1, 0, 1]) + '\n',
        'old2.txt': '2\n16\n16\n' + '\n'.join(['1'] * 16) + '\n',
    }
    for name, text in inputs.items():
        with open(os.path.join(tmp, name), 'w') as fh:
            fh.write(text)

    def describe(path, sort_counts=False):
        with h5py.File(path, 'r') as f:
            root = dict((k, v.decode() if isinstance(v, bytes) else str(v)) for k, v in f.attrs.items())
            profiles = {}
            for name in sorted(f['profiles']):
                ds = f['profiles/' + name]
                counts = ds[:].astype('<i8')
                if sort_counts:
                    counts = np.sort(counts)
                profiles[name] = {'attrs': dict((k, float(v) if isinstance(v, (float, np.floating)) else int(v))
                                                for k, v in ds.attrs.items()),
                                  'sha256': hashlib.sha256(counts.tobytes()).hexdigest()}
                if sort_counts:      # the attributes that do not depend on the order
                    profiles[name]['attrs'] = dict((k, v) for k, v in profiles[name]['attrs'].items())
        return {'root': root, 'profiles': profiles}

    steps = [
        ['count', '-k', '8', 'a_1.fa', 'a_2.fa', 'b_1.fa', 'b_2.fa', 'c_1.fa', 'c_2.fa', 'counted.k8'],
        ['count', '-k', '5', 'a_1.fa', 'a_2.fa', 'named.k5', '-p', 'x', 'y'],
        ['count', '-k', '4', '--by-record', 'rec.fa', 'rec.k4'],
        ['count', '-k', '3', '-r', 'rec.fa', 'rec.fa', 'rec2.k3', '-p', 'p', 'q'],
        ['count', 'a_1.fa', 'default.k9'],
        ['info', 'counted.k8'],
        ['info', 'counted.k8', '-p', 'b_2'],
        ['stats', 'counted.k8'],
        ['stats', '-n', '4', 'counted.k8', '-p', 'b_2', 'a_1'],
        ['showbalance', 'counted.k8'],
        ['showbalance', '-n', '3', 'counted.k8', '-p', 'b_1'],
        ['distr', 'counted.k8', 'distr.txt', '-p', 'a_1', 'c_2'],
        ['getcount', 'counted.k8', 'ACGTACGT'],
        ['getcount', 'counted.k8', 'TTTTTTTT', '-p', 'c_1'],
        ['merge', 'counted.k8', 'counted.k8', 'merged.k8', '-l', 'a_1', 'b_1', '-r', 'a_2', 'b_2'],
        ['merge', '-m', 'xor', 'counted.k8', 'counted.k8', 'merged_xor.k8', '-l', 'a_1', '-r', 'a_2'],
        ['merge', '-m', 'nint', 'counted.k8', 'counted.k8', 'merged_nint.k8', '-l', 'c_1', '-r', 'c_1'],
        ['merge', '-c', 'np.maximum(left, right)', 'counted.k8', 'counted.k8', 'merged_custom.k8', '-l', 'c_1', '-r', 'c_2'],
        ['balance', 'counted.k8', 'balanced.k8', '-p', 'a_1', 'c_2'],
        ['cat', 'counted.k8', 'merged.k8', 'cat.k8', '-x', 'p_', 'q_'],
        ['cat', 'counted.k8', 'merged.k8', 'cat_sel.k8', '-p', 'a_1', 'b_1_b_2', 'nosuch'],
        ['cat', 'counted.k8', 'left.k8', '-p', 'a_1', 'b_1', 'c_1'],
        ['cat', 'counted.k8', 'right.k8', '-p', 'a_2', 'b_2', 'c_2'],
        ['positive', 'left.k8', 'right.k8', 'pos_l.k8', 'pos_r.k8'],
        ['scale', 'left.k8', 'right.k8', 'sc_l.k8', 'sc_r.k8'],
        ['scale', '-d', 'left.k8', 'right.k8', 'scd_l.k8', 'scd_r.k8', '-l', 'a_1', '-r', 'c_2'],
        ['shrink', 'counted.k8', 'shrunk.k7'],
        ['shrink', '-f', '3', 'counted.k8', 'shrunk.k5', '-p', 'c_1'],
        ['shuffle', 'counted.k8', 'shuffled.k8', '-p', 'a_1'],
        ['smooth', 'left.k8', 'right.k8', 'sm_l.k8', 'sm_r.k8', '-s', 'average', '-t', '2'],
        ['smooth', 'left.k8', 'right.k8', 'smc_l.k8', 'smc_r.k8', '-M', 'np.max(values)', '-t', '3', '-l', 'a_1', '-r', 'b_2'],
        ['distance', 'left.k8', 'right.k8'],
        ['distance', '-n', '3', 'left.k8', 'right.k8'],
        ['distance', '-b', '-n', '8', 'left.k8', 'right.k8'],
        ['distance', '-P', 'sum', '-n', '8', 'left.k8', 'right.k8'],
        ['distance', '-D', 'euclidean', '-n', '6', 'left.k8', 'right.k8'],
        ['distance', '-D', 'cosine', '-n', '8', 'left.k8', 'right.k8'],
        ['distance', '-m', '-s', 'median', '-t', '1', '-S', '-d', '-n', '8', 'left.k8', 'right.k8'],
        ['distance', '--positive', '-S', '-n', '8', 'left.k8', 'right.k8'],
        ['distance', '-f', 'abs(left - right) / (left + right + 2)', '-n', '8', 'left.k8', 'right.k8'],
        ['distance', '-n', '8', 'left.k8', 'right.k8', '-l', 'c_1', 'a_1', '-r', 'b_2', 'b_2'],
        ['matrix', 'counted.k8', 'matrix.txt', '-n', '3'],
        ['matrix', '-b', '-n', '8', 'counted.k8', 'matrix_b.txt', '-p', 'c_2', 'a_1', 'b_1'],
        ['matrix', '-D', 'euclidean', '-n', '6', 'counted.k8', 'matrix_e.txt'],
        ['matrix', '-m', '-s', 'min', '-t', '1', '-S', '-n', '8', 'counted.k8', 'matrix_s.txt'],
        ['convert', 'old1.txt', 'old2.txt', 'converted.k2'],
        ['convert', 'old1.txt', 'converted_named.k2', '-p', 'o1'],
        # usage errors (exit status 2)
        ['count', '-k', '8', 'a_1.fa', 'counted.k8'],                              # the output exists
        ['info', 'a_1.fa'],                                                         # not a k-mer profile file
        ['info', 'nosuch.k8'],
        ['getcount', 'counted.k8', 'ACGT'],
        ['getcount', 'counted.k8', 'ACGTNCGT'],
        ['matrix', 'counted.k8', 'matrix_one.txt', '-p', 'a_1'],
        ['distance', 'left.k8', 'right.k8', '-l', 'a_1'],
        ['distance', 'left.k8', 'named.k5', '-l', 'a_1', '-r', 'x'],
        ['count', '-k', '4', 'a_1.fa', 'a_2.fa', 'bad_names.k4', '-p', 'only_one'],
        ['cat', 'counted.k8', 'merged.k8', 'bad_prefix.k8', '-x', 'p_'],
        ['shrink', '-f', '8', 'counted.k8', 'bad_shrink.k0'],
        ['nosuchcommand'],
    ]
    out = []
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        for argv in steps:
            before = set(os.listdir(tmp))
            so, se = io.StringIO(), io.StringIO()
            status = 0
            with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
                try:
                    kmer.main(argv)
                except SystemExit as e:
                    status = e.code
            rec = {'argv': argv, 'status': status, 'stdout': so.getvalue()}
            if status:
                err = se.getvalue().strip().split('\n')[-1]
                rec['error'] = err.split('error: ', 1)[1] if 'error: ' in err else err
            created = sorted(set(os.listdir(tmp)) - before)
            rec['text_files'] = {}
            rec['profile_files'] = {}
            for name in created:
                path = os.path.join(tmp, name)
                if name.endswith('.txt'):
                    with open(path) as fh:
                        rec['text_files'][name] = fh.read()
                elif status == 0:
                    rec['profile_files'][name] = describe(path, sort_counts=argv[0] == 'shuffle')
                else:
                    rec['left_behind'] = rec.get('left_behind', []) + [name]
            out.append(rec)
    finally:
        os.chdir(cwd)
    return {'inputs': inputs, 'steps': out}


def main():
    os.makedirs(OUT, exist_ok=True)
    arrays = {}
    meta = {'generator': 'tools/gen_golden.py', 'reference': 'kPAL 2.1.2.dev (/root/reference)',
            'python': sys.version.split()[0], 'numpy': np.__version__}
    if sys.argv[1:] == ['g12']:
        with open(os.path.join(OUT, 'cli.json'), 'w') as fh:
            json.dump({'meta': meta, 'G12': g12()}, fh, indent=0)
        print('wrote cli.json')
        return
    with open(os.path.join(OUT, 'counts.json'), 'w') as fh:
        json.dump({'meta': meta, 'G1': g1(), 'G2': g2()}, fh)
    with open(os.path.join(OUT, 'synth.json'), 'w') as fh:
        json.dump({'meta': meta, 'G3': g3()}, fh, indent=1)
    scal = {'meta': meta, 'G4': g4(), 'G5': g5(arrays), 'G6': g6(), 'G7': g7(arrays), 'G8': g8()}
    with open(os.path.join(OUT, 'scalars.json'), 'w') as fh:
        json.dump(scal, fh, indent=1)
    np.savez_compressed(os.path.join(OUT, 'vectors.npz'), **arrays)
    opt_arrays = {}
    with open(os.path.join(OUT, 'options.json'), 'w') as fh:
        json.dump({'meta': meta, 'G9': g9(opt_arrays)}, fh, indent=0)
    np.savez_compressed(os.path.join(OUT, 'options.npz'), **opt_arrays)
    sum_arrays = {}
    with open(os.path.join(OUT, 'summaries.json'), 'w') as fh:
        json.dump({'meta': meta, 'G10': g10(sum_arrays)}, fh, indent=0)
    np.savez_compressed(os.path.join(OUT, 'summaries.npz'), **sum_arrays)
    with open(os.path.join(OUT, 'callers.json'), 'w') as fh:
        json.dump({'meta': meta, 'G11': g11()}, fh, indent=0)
    with open(os.path.join(OUT, 'cli.json'), 'w') as fh:
        json.dump({'meta': meta, 'G12': g12()}, fh, indent=0)
    print('wrote', sorted(os.listdir(OUT)))


if __name__ == '__main__':
    main()
