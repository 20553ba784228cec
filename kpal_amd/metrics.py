"""Drop-in for ``kpal.metrics`` with the multiset / euclidean reductions on the GPU.

API parity: function names, argument order and the keys of the ``pairwise``,
``vector_distance``, ``summary`` and ``mergers`` dicts are those of the reference
(kpal/metrics.py; the dict keys are the CLI ``choices``, kpal/kmer.py:776,812,816,879).

Routing rule: ``multiset`` with the built-in ``pairwise['prod']`` /
``pairwise['sum']`` objects (recognised by identity) and ``euclidean`` on integer vectors run
as fused HIP reductions; a user-supplied pairwise callable cannot enter a kernel and keeps the
reference's NumPy formulation.  The built-in mergers run as one HIP kernel through ``Profile.merge``
(``merger_code``); the scale-factor helpers are scalar arithmetic and stay NumPy.
"""
from collections import Counter

import numpy as np

from . import _native


def distribution(vector):
    """Sorted ``(value, count)`` pairs of ``vector`` (kpal/metrics.py:22-33)."""
    return sorted(Counter(vector).items())


def vector_length(vector):
    """Euclidean norm, ``sqrt(dot(v, v))`` (kpal/metrics.py:36-46).

    For integer vectors the dot product is the exact (wrapping) int64 the reference gets from
    ``np.dot`` -- computed on the GPU -- followed by one IEEE sqrt.
    """
    v = np.asanyarray(vector)
    if v.dtype.kind in 'iub' and v.ndim == 1:
        return np.float64(_native.context().pair_distance(v, np.zeros_like(v, dtype=np.int64), _native.EUCLIDEAN))
    return np.sqrt(np.dot(v, v))


def get_scale(left, right):
    """Scale factors from the totals; one of them is 1.0 (kpal/metrics.py:49-72)."""
    total_left = np.sum(left)
    total_right = np.sum(right)
    if total_left < total_right:
        return total_right / total_left, 1.0
    return 1.0, total_left / total_right


def scale_down(left, right):
    """Normalise two scale factors by the larger one (kpal/metrics.py:75-86)."""
    top = max(left, right)
    return left / top, right / top


def positive(vector, mask):
    """Zero ``vector`` wherever ``mask`` is zero (kpal/metrics.py:89-98)."""
    return np.multiply(vector, np.asanyarray(mask, dtype=bool))


def _pairwise_prod(x, y):
    # f(x, y) = |x - y| / ((x + 1)(y + 1))          kpal/metrics.py:160, doc/method.rst:68-71
    return abs(x - y) / ((x + 1) * (y + 1))


def _pairwise_sum(x, y):
    # f(x, y) = |x - y| / (x + y + 1)               kpal/metrics.py:161, doc/method.rst:73-76
    return abs(x - y) / (x + y + 1)


def multiset(left, right, pairwise):
    """Multiset distance (kpal/metrics.py:101-123, doc/method.rst:63-66).

    ``sum(f(l_i, r_i) for i where l_i or r_i) / (#such i + 1)``.  With a built-in pairwise
    function this is one fused HIP pass (mask, pairwise term, fp64 sum and the non-zero count in a
    single read of both vectors).
    """
    left = np.asanyarray(left)
    right = np.asanyarray(right)
    code = _PAIRWISE_CODE.get(id(pairwise))
    if code is not None and left.ndim == 1 and left.shape == right.shape:
        if left.dtype.kind in 'iub' and right.dtype.kind in 'iub':
            return _native.context().pair_distance(left, right, code)
        if left.dtype.kind in 'fiub' and right.dtype.kind in 'fiub':
            return _native.context().pair_distance_f64(left, right, code)
    # user-defined vectorised pairwise function: reference formulation
    keep = np.where(np.logical_or(left, right))
    terms = pairwise(left[keep], right[keep])
    return terms.sum() / (len(terms) + 1)


def euclidean(left, right):
    """Euclidean distance (kpal/metrics.py:126-135); exact int64 dot product on the GPU."""
    l = np.asanyarray(left)
    r = np.asanyarray(right)
    if l.dtype.kind in 'iub' and r.dtype.kind in 'iub' and l.ndim == 1 and l.shape == r.shape:
        return np.float64(_native.context().pair_distance(l, r, _native.EUCLIDEAN))
    d = np.subtract(l, r)
    return np.sqrt(np.dot(d, d))


def cosine_similarity(left, right):
    """Cosine similarity (kpal/metrics.py:138-147)."""
    return np.dot(left, right) / (vector_length(left) * vector_length(right))


#: Vector distance functions (keys: kpal/metrics.py:151-155).
vector_distance = {
    'default': None,
    'euclidean': euclidean,
    'cosine': cosine_similarity,
}

#: Pairwise distance functions on numpy arrays (keys: kpal/metrics.py:159-162).
pairwise = {
    'prod': _pairwise_prod,
    'sum': _pairwise_sum,
}

_PAIRWISE_CODE = {id(_pairwise_prod): _native.PAIRWISE_PROD, id(_pairwise_sum): _native.PAIRWISE_SUM}

#: Summary functions (keys: kpal/metrics.py:166-170).
summary = {
    'min': np.min,
    'average': np.mean,
    'median': np.median,
}

#: Merge functions on numpy arrays (keys: kpal/metrics.py:174-179).
mergers = {
    'sum': lambda x, y: x + y,
    'xor': lambda x, y: (x + y) * np.logical_xor(x, y),
    'int': lambda x, y: x * np.asanyarray(y, dtype=bool),
    'nint': lambda x, y: x * np.logical_not(y),
}


def merger_code(function):
    """-> native code of a built-in merge function (a value of :data:`mergers`), else None."""
    for code, name in ((_native.MERGE_SUM, 'sum'), (_native.MERGE_XOR, 'xor'), (_native.MERGE_INT, 'int'),
                       (_native.MERGE_NINT, 'nint')):
        if function is mergers[name]:
            return code
    return None


def summary_code(function):
    """-> native code of a built-in summary function (``np.min`` / ``np.mean`` / ``np.median``,
    the values of :data:`summary`), else None."""
    for code, builtin in ((_native.SUMMARY_MIN, np.min), (_native.SUMMARY_AVERAGE, np.mean),
                          (_native.SUMMARY_MEDIAN, np.median)):
        if function is builtin:
            return code
    return None


def pairwise_code(function):
    """-> native metric code for a built-in pairwise function, else None."""
    return _PAIRWISE_CODE.get(id(function))
