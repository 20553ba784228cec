"""ctypes binding of oracle/libkpal_oracle.so (TEST INFRASTRUCTURE ONLY).

Parity pinned: checked against reference-generated goldens in tests/test_oracle_golden.py.
Function names mirror the reference symbols they restate (kpal/klib.py, kpal/metrics.py,
kpal/kdistlib.py); citations live in kpal_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_c_i64p = ctypes.POINTER(ctypes.c_int64)
_c_f64p = ctypes.POINTER(ctypes.c_double)
_c_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force=False):
    """Compile libkpal_oracle.so with gcc (building the checker is not using it)."""
    so = os.path.join(_HERE, 'libkpal_oracle.so')
    src = os.path.join(_HERE, 'kpal_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'libkpal_oracle.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.kpal_oracle_count_sequence.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, _c_i64p]
        L.kpal_oracle_count_sequence.restype = ctypes.c_int
        L.kpal_oracle_count_piece.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                              ctypes.c_int, _c_i64p]
        L.kpal_oracle_count_piece.restype = ctypes.c_int
        L.kpal_oracle_count_flat_mt.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _c_i64p]
        L.kpal_oracle_count_flat_mt.restype = ctypes.c_int
        L.kpal_oracle_count_flat_mt_mode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i64p]
        L.kpal_oracle_count_flat_mt_mode.restype = ctypes.c_int
        L.kpal_oracle_count_blocks_mt.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                  ctypes.c_int, ctypes.c_int, _c_i64p, _c_i64p]
        L.kpal_oracle_count_blocks_mt.restype = ctypes.c_int
        L.kpal_oracle_reverse_complement.argtypes = [ctypes.c_uint64, ctypes.c_int]
        L.kpal_oracle_reverse_complement.restype = ctypes.c_uint64
        L.kpal_oracle_balance.argtypes = [_c_i64p, ctypes.c_int]
        L.kpal_oracle_balance.restype = None
        L.kpal_oracle_balance_mt.argtypes = [_c_i64p, ctypes.c_int, ctypes.c_int]
        L.kpal_oracle_balance_mt.restype = None
        L.kpal_oracle_split.argtypes = [_c_i64p, ctypes.c_int, _c_i64p, _c_i64p]
        L.kpal_oracle_split.restype = ctypes.c_size_t
        L.kpal_oracle_multiset_i64.argtypes = [_c_i64p, _c_i64p, ctypes.c_size_t, ctypes.c_int, _c_i64p]
        L.kpal_oracle_multiset_i64.restype = ctypes.c_double
        L.kpal_oracle_multiset_f64.argtypes = [_c_f64p, _c_f64p, ctypes.c_size_t, ctypes.c_int, _c_i64p]
        L.kpal_oracle_multiset_f64.restype = ctypes.c_double
        L.kpal_oracle_euclidean_i64.argtypes = [_c_i64p, _c_i64p, ctypes.c_size_t, _c_i64p]
        L.kpal_oracle_euclidean_i64.restype = ctypes.c_double
        L.kpal_oracle_distance.argtypes = [_c_i64p, _c_i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.kpal_oracle_distance.restype = ctypes.c_double
        L.kpal_oracle_distance_matrix.argtypes = [_c_i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, _c_f64p]
        L.kpal_oracle_distance_matrix.restype = None
        L.kpal_oracle_positive.argtypes = [_c_i64p, _c_i64p, ctypes.c_size_t]
        L.kpal_oracle_positive.restype = None
        L.kpal_oracle_dynamic_smooth.argtypes = [_c_i64p, _c_i64p, ctypes.c_int, ctypes.c_int, ctypes.c_double]
        L.kpal_oracle_dynamic_smooth.restype = None
        L.kpal_oracle_profile_distance.argtypes = [_c_i64p, _c_i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_int]
        L.kpal_oracle_profile_distance.restype = ctypes.c_double
        L.kpal_oracle_stats.argtypes = [_c_i64p, ctypes.c_size_t, _c_i64p, _c_f64p]
        L.kpal_oracle_stats.restype = None
        L.kpal_oracle_merge.argtypes = [_c_i64p, _c_i64p, ctypes.c_size_t, ctypes.c_int, _c_i64p]
        L.kpal_oracle_merge.restype = None
        L.kpal_oracle_shrink.argtypes = [_c_i64p, ctypes.c_int, ctypes.c_int, _c_i64p]
        L.kpal_oracle_shrink.restype = None
        L.kpal_oracle_strand_balance.argtypes = [_c_i64p, ctypes.c_int, ctypes.c_int]
        L.kpal_oracle_strand_balance.restype = ctypes.c_double
        L.kpal_oracle_synth_reads.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.kpal_oracle_synth_reads.restype = None
        _LIB = L
    return _LIB


PAIRWISE = {'prod': 0, 'sum': 1}
METRIC = {'prod': 0, 'sum': 1, 'euclidean': 2, 'cosine': 3}
SUMMARY = {'min': 0, 'average': 1, 'median': 2}


def _i64(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_c_i64p)


def _bytes_view(seq):
    if isinstance(seq, str):
        seq = seq.encode('latin-1', 'replace')
    if isinstance(seq, (bytes, bytearray, memoryview)):
        return np.frombuffer(seq, dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


def from_sequences(sequences, length):
    """Restates Profile.from_sequences (klib.py:135-170) -> int64[4**length]."""
    counts = np.zeros(4 ** length, dtype=np.int64)
    cp = counts.ctypes.data_as(_c_i64p)
    L = lib()
    for s in sequences:
        b = _bytes_view(s)
        if b.size:
            rc = L.kpal_oracle_count_sequence(b.ctypes.data, b.size, length, cp)
            if rc:
                raise ValueError('bad k')
    return counts


COUNT_MODES = {'auto': 0, 'shared': 1, 'private': 2}


def count_flat(buf, length, threads=1, mode='auto'):
    """Count a flat byte stream (any non-AaCcGgTt byte separates); optionally N threads (the all-cores
    baseline / at-scale oracle).  mode: 'shared' = one table, relaxed atomic adds; 'private' = one table per
    thread + integer merge; 'auto' = private for k <= 8, else shared.  Same result in every mode."""
    b = _bytes_view(buf)
    L = lib()
    n = b.size
    if threads <= 1 or n < (1 << 16):
        counts = np.zeros(4 ** length, dtype=np.int64)
        if n:
            L.kpal_oracle_count_piece(b.ctypes.data, 0, n, length, counts.ctypes.data_as(_c_i64p))
        return counts
    counts = np.zeros(4 ** length, dtype=np.int64)
    rc = L.kpal_oracle_count_flat_mt_mode(b.ctypes.data, n, length, int(threads), COUNT_MODES[mode], counts.ctypes.data_as(_c_i64p))
    if rc:
        raise MemoryError('kpal_oracle_count_flat_mt_mode failed (%d)' % rc)
    return counts


def count_blocks(buf, length, block_bits, blocks, threads=1):
    """Counts of SELECTED blocks of ``2**block_bits`` consecutive table entries of a flat byte stream, for tables too large
    to hold on the host (k = 15: 8 GiB): -> (plain, mirror), int64[len(blocks), 2**block_bits] each, with
    ``plain[s, j] = counts[blocks[s] * 2**block_bits + j]`` and ``mirror[s, j] = counts[rc(blocks[s] * 2**block_bits + j)]``,
    so that ``plain + mirror`` is ``Profile.balance`` (klib.py:285-298) of the selected entries."""
    b = _bytes_view(buf)
    sel = np.ascontiguousarray(blocks, dtype=np.uint64)
    plain = np.zeros((sel.size, 1 << block_bits), dtype=np.int64)
    mirror = np.zeros_like(plain)
    rc = lib().kpal_oracle_count_blocks_mt(b.ctypes.data, b.size, int(length), int(block_bits), sel.ctypes.data, int(sel.size),
                                           int(max(threads, 1)), plain.ctypes.data_as(_c_i64p), mirror.ctypes.data_as(_c_i64p))
    if rc:
        raise ValueError('kpal_oracle_count_blocks_mt failed (%d)' % rc)
    return plain, mirror


def reverse_complement(number, length):
    return int(lib().kpal_oracle_reverse_complement(number, length))


def balance(counts, length, threads=None):
    """Restates Profile.balance (klib.py:285-298); returns a balanced COPY.  Tables of k >= 11 are dealt to host threads by
    index range (the same statement per pair; ``threads=1`` is the plain loop)."""
    c = np.array(counts, dtype=np.int64, copy=True)
    if threads is None:
        threads = min(16, os.cpu_count() or 1) if length >= 11 else 1
    if threads > 1:
        lib().kpal_oracle_balance_mt(c.ctypes.data_as(_c_i64p), length, threads)
    else:
        lib().kpal_oracle_balance(c.ctypes.data_as(_c_i64p), length)
    return c


def split(counts, length):
    """Restates Profile.split (klib.py:300-327)."""
    c, cp = _i64(counts)
    f = np.empty(c.size, dtype=np.int64)
    r = np.empty(c.size, dtype=np.int64)
    m = lib().kpal_oracle_split(cp, length, f.ctypes.data_as(_c_i64p), r.ctypes.data_as(_c_i64p))
    return f[:m].copy(), r[:m].copy()


def multiset(left, right, pairwise='prod', return_m=False):
    """Restates metrics.multiset (metrics.py:101-123) with the built-in pairwise functions."""
    left = np.asanyarray(left)
    right = np.asanyarray(right)
    m = ctypes.c_int64(0)
    if left.dtype.kind == 'f' or right.dtype.kind == 'f':
        l = np.ascontiguousarray(left, dtype=np.float64)
        r = np.ascontiguousarray(right, dtype=np.float64)
        d = lib().kpal_oracle_multiset_f64(l.ctypes.data_as(_c_f64p), r.ctypes.data_as(_c_f64p),
                                           l.size, PAIRWISE[pairwise], ctypes.byref(m))
    else:
        l, lp = _i64(left)
        r, rp = _i64(right)
        d = lib().kpal_oracle_multiset_i64(lp, rp, l.size, PAIRWISE[pairwise], ctypes.byref(m))
    return (d, m.value) if return_m else d


def euclidean(left, right, return_dot=False):
    """Restates metrics.euclidean (metrics.py:126-135) for int64 vectors."""
    l, lp = _i64(left)
    r, rp = _i64(right)
    dot = ctypes.c_int64(0)
    d = lib().kpal_oracle_euclidean_i64(lp, rp, l.size, ctypes.byref(dot))
    return (d, dot.value) if return_dot else d


def distance(left, right, length, do_balance=False, metric='prod'):
    """Restates ProfileDistance.distance default/do_balance paths (kdistlib.py:126-161)."""
    l, lp = _i64(left)
    r, rp = _i64(right)
    return lib().kpal_oracle_distance(lp, rp, length, int(do_balance), METRIC[metric])


def dynamic_smooth(left, right, length, summary='min', threshold=0):
    """Restates ProfileDistance.dynamic_smooth (kdistlib.py:53-124); returns smoothed copies."""
    l = np.array(left, dtype=np.int64)
    r = np.array(right, dtype=np.int64)
    lib().kpal_oracle_dynamic_smooth(l.ctypes.data_as(_c_i64p), r.ctypes.data_as(_c_i64p), length,
                                     SUMMARY[summary], float(threshold))
    return l, r


def profile_distance(left, right, length, do_balance=False, do_positive=False, do_smooth=False, summary='min',
                     threshold=0, do_scale=False, down=False, metric='prod'):
    """Restates ProfileDistance.distance with every option (kdistlib.py:126-161)."""
    l, lp = _i64(left)
    r, rp = _i64(right)
    return lib().kpal_oracle_profile_distance(lp, rp, length, int(do_balance), int(do_positive), int(do_smooth),
                                              SUMMARY[summary], float(threshold), int(do_scale), int(down),
                                              METRIC[metric])


def distance_matrix_values(profiles, length, do_balance=False, metric='prod', threads=1):
    """Lower-triangle values in kdistlib.distance_matrix order (kdistlib.py:179-186).  ``threads`` > 1: the pairs are dealt
    to that many threads (every value still comes from the single-threaded pair function)."""
    p = profiles if isinstance(profiles, np.ndarray) and profiles.ndim == 2 else np.stack([np.asarray(x, dtype=np.int64) for x in profiles])
    p = np.ascontiguousarray(p, dtype=np.int64)
    P = p.shape[0]
    out = np.empty(P * (P - 1) // 2, dtype=np.float64)
    if threads > 1:
        L = lib()
        L.kpal_oracle_distance_matrix_mt.argtypes = [_c_i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_f64p]
        L.kpal_oracle_distance_matrix_mt.restype = ctypes.c_int
        rc = L.kpal_oracle_distance_matrix_mt(p.ctypes.data_as(_c_i64p), P, length, int(do_balance), METRIC[metric], int(threads),
                                              out.ctypes.data_as(_c_f64p))
        if rc:
            raise MemoryError('kpal_oracle_distance_matrix_mt failed (%d)' % rc)
        return out
    lib().kpal_oracle_distance_matrix(p.ctypes.data_as(_c_i64p), P, length, int(do_balance),
                                      METRIC[metric], out.ctypes.data_as(_c_f64p))
    return out


def distance_matrix_text(names, values, precision):
    """Text layout of kdistlib.distance_matrix (kdistlib.py:176-186)."""
    n = len(names)
    lines = [str(n)] + list(names)
    o = 0
    for i in range(1, n):
        lines.append(' '.join('{{0:.{0}f}}'.format(precision).format(values[o + j]) for j in range(i)))
        o += i
    return '\n'.join(lines) + '\n'


MERGER = {'sum': 0, 'xor': 1, 'int': 2, 'nint': 3}


def stats(counts):
    """Restates Profile.total/non_zero/mean/median/std (klib.py:193-225) -> dict (plus min, max)."""
    c, cp = _i64(counts)
    iout = np.empty(4, dtype=np.int64)
    dout = np.empty(3, dtype=np.float64)
    lib().kpal_oracle_stats(cp, c.size, iout.ctypes.data_as(_c_i64p), dout.ctypes.data_as(_c_f64p))
    return {'total': int(iout[0]), 'non_zero': int(iout[1]), 'min': int(iout[2]), 'max': int(iout[3]),
            'mean': float(dout[0]), 'median': float(dout[1]), 'std': float(dout[2])}


def merge(left, right, merger='sum'):
    """Restates metrics.mergers[merger](left, right) on int64 vectors (metrics.py:174-179)."""
    l, lp = _i64(left)
    r, rp = _i64(right)
    out = np.empty(l.size, dtype=np.int64)
    lib().kpal_oracle_merge(lp, rp, l.size, MERGER[merger], out.ctypes.data_as(_c_i64p))
    return out


def shrink(counts, length, factor=1):
    """Restates Profile.shrink (klib.py:329-352) -> the 4^(length-factor) new counts."""
    c, cp = _i64(counts)
    out = np.empty(4 ** (length - factor), dtype=np.int64)
    lib().kpal_oracle_shrink(cp, length, factor, out.ctypes.data_as(_c_i64p))
    return out


def strand_balance(counts, length, pairwise='prod'):
    """Restates kmer.get_balance's score (kmer.py:243-245)."""
    c, cp = _i64(counts)
    return lib().kpal_oracle_strand_balance(cp, length, PAIRWISE[pairwise])


def synth_reads(seed, first_read, n_reads, read_len=150, noisy=False):
    """SURVEY.md 8d generator -> uint8[n_reads*(read_len+1)] ('\\n'-terminated reads)."""
    out = np.empty(n_reads * (read_len + 1), dtype=np.uint8)
    lib().kpal_oracle_synth_reads(seed, first_read, n_reads, read_len, int(noisy), out.ctypes.data)
    return out
