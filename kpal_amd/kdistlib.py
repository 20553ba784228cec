"""Drop-in for ``kpal.kdistlib``: :class:`ProfileDistance` and :func:`distance_matrix`.

The default pipeline of ``ProfileDistance.distance`` -- copy, optional balance, multiset or
euclidean (kpal/kdistlib.py:126-161) -- runs as fused HIP kernels (``kpal_pair_distance``), and
``distance_matrix`` over P profiles is ONE tiled kernel launch (``kpal_distance_matrix``) instead
of P(P-1)/2 Python-level distances; profiles are balanced once each, which is identical to the
reference balancing copies inside every pair.  The optional positive / dynamic-smooth / scale
steps (kpal/kdistlib.py:143-157) and cosine similarity run on the device too
(``kpal_profile_distance``) whenever every callable is one of kPAL's built-ins (the values of
``metrics.summary`` / ``metrics.pairwise`` / ``metrics.vector_distance``, recognised by identity);
a user-supplied callable cannot enter a kernel and keeps the reference's NumPy formulation.
"""
import numpy as np

from . import _native, metrics


def _plain_int64(a):
    return (isinstance(a, np.ndarray) and a.dtype == np.int64 and a.ndim == 1 and a.flags['C_CONTIGUOUS']
            and a.flags['WRITEABLE'])


def _is_number(x):
    return isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, bool)


class ProfileDistance(object):
    """Configurable distance between two profiles (kpal/kdistlib.py:21-51)."""

    def __init__(self, do_balance=False, do_positive=False, do_smooth=False,
                 summary=metrics.summary['min'], threshold=0, do_scale=False,
                 down=False, distance_function=None,
                 pairwise=metrics.pairwise['prod']):
        self._do_balance = do_balance
        self._do_positive = do_positive
        self._do_smooth = do_smooth
        self._threshold = threshold
        self._do_scale = do_scale
        self._down = down
        self._distance_function = distance_function
        self._pairwise = pairwise
        self._function = summary

    # ---- dynamic smoothing (kpal/kdistlib.py:53-124) ---------------------------------------------
    def _collapse(self, vector, start, length):
        """Sums of the four quarters of ``vector[start:start+length]``."""
        return np.reshape(vector[start:start + length], (4, length // 4)).sum(axis=1)

    def _dynamic_smooth(self, left, right, start, length):
        if length == 1:
            return
        left_c = self._collapse(left.counts, start, length)
        right_c = self._collapse(right.counts, start, length)
        if min(self._function(left_c), self._function(right_c)) <= self._threshold:
            left.counts[start] = left_c.sum()
            right.counts[start] = right_c.sum()
            left.counts[start + 1:start + length] = 0
            right.counts[start + 1:start + length] = 0
            return
        quarter = length // 4
        for i in range(4):
            self._dynamic_smooth(left, right, start + i * quarter, quarter)

    def dynamic_smooth(self, left, right):
        """Collapse sub-profiles that fail the summary/threshold test, in place
        (kpal/kdistlib.py:112-124).  Built-in summary functions on int64 profiles run as a
        level-wise tree reduction on the GPU; anything else takes the reference's recursion."""
        code = metrics.summary_code(self._function)
        lc, rc = left.counts, right.counts
        if (code is not None and _plain_int64(lc) and _plain_int64(rc) and lc.size == rc.size
                and _is_number(self._threshold)):
            _native.context().dynamic_smooth(lc, rc, left.length, code, self._threshold)
            return
        self._dynamic_smooth(left, right, 0, left.number)

    # ---- routing -------------------------------------------------------------------------------
    def _native_metric(self):
        """Metric code when the final reduction can run on the GPU, else None."""
        if not self._distance_function:
            return metrics.pairwise_code(self._pairwise)
        if self._distance_function is metrics.euclidean:
            return _native.EUCLIDEAN
        if self._distance_function is metrics.cosine_similarity:
            return _native.COSINE
        return None

    def _native_options(self):
        """``kpal_distance_options`` for this configuration, or None if a user-supplied callable
        or a non-numeric threshold keeps it in Python."""
        metric = self._native_metric()
        if metric is None:
            return None
        summary = 0
        if self._do_smooth:
            summary = metrics.summary_code(self._function)
            if summary is None or not _is_number(self._threshold):
                return None
        return _native.DistanceOptions(
            do_balance=int(bool(self._do_balance)), do_positive=int(bool(self._do_positive)),
            do_smooth=int(bool(self._do_smooth)), summary=summary,
            threshold=float(self._threshold) if self._do_smooth else 0.0,
            do_scale=int(bool(self._do_scale)), down=int(bool(self._down)), metric=metric)

    def _is_plain(self):
        return not (self._do_positive or self._do_smooth or self._do_scale)

    def distance(self, left, right):
        """Distance between two profiles; the inputs are left unmodified
        (kpal/kdistlib.py:126-161, tests/test_kdistlib.py:124-135)."""
        metric = self._native_metric()
        dev = _device_pair(left, right)
        if dev is not None:
            # both tables are still in HBM (klib.Profile.from_fasta_by_record): the same kernels on the device copies, no transfer
            ctx, dl, dr = dev
            if self._is_plain() and metric is not None and metric != _native.COSINE:
                return ctx.pair_distance_device(4 ** left.length, dl, dr, metric, do_balance=self._do_balance, k=left.length)
            options = self._native_options()
            if options is not None:
                return ctx.profile_distance_device(left.length, dl, dr, options)
        integer = (np.asanyarray(left.counts).dtype.kind in 'iub' and np.asanyarray(right.counts).dtype.kind in 'iub'
                   and len(left.counts) == len(right.counts))
        if self._is_plain() and metric is not None and metric != _native.COSINE and integer:
            # fused: balanced copies are made on the device, nothing is written back
            return _native.context().pair_distance(left.counts, right.counts, metric,
                                                   do_balance=self._do_balance, k=left.length)
        options = self._native_options() if integer else None
        if options is not None:
            # the whole option pipeline on device copies (kdistlib.py:136-161)
            return _native.context().profile_distance(left.counts, right.counts, left.length, options)

        left = left.copy()
        right = right.copy()
        if self._do_balance:
            left.balance()
            right.balance()
        if self._do_positive:
            left.counts = metrics.positive(left.counts, right.counts)
            right.counts = metrics.positive(right.counts, left.counts)
        if self._do_smooth:
            self.dynamic_smooth(left, right)
        if self._do_scale:
            left_scale, right_scale = metrics.get_scale(left.counts, right.counts)
            if self._down:
                left_scale, right_scale = metrics.scale_down(left_scale, right_scale)
            left.counts = left.counts * left_scale
            right.counts = right.counts * right_scale
        if not self._distance_function:
            return metrics.multiset(left.counts, right.counts, self._pairwise)
        return self._distance_function(left.counts, right.counts)


def distance_matrix(profiles, output, precision, dist):
    """Write the lower-triangular distance matrix of ``profiles`` to ``output``
    (kpal/kdistlib.py:164-186): the count, the names, then row i = distances to profiles
    0..i-1, ``precision`` decimals, space separated."""
    count = len(profiles)
    print(str(count), file=output)
    for profile in profiles:
        print(profile.name, file=output)
    if count < 2:
        return

    metric = dist._native_metric()
    same_k = len(set(p.length for p in profiles)) == 1
    values = _device_matrix(profiles, dist, metric) if same_k else None
    if values is not None:
        _write_matrix(output, count, values, precision)
        return
    integer = all(np.asanyarray(p.counts).dtype.kind in 'iub' for p in profiles)
    options = dist._native_options() if (same_k and integer) else None
    if dist._is_plain() and metric is not None and metric != _native.COSINE and same_k and integer:
        values = _native.context().distance_matrix([p.counts for p in profiles], profiles[0].length, metric,
                                                   do_balance=dist._do_balance)
    elif options is not None:
        # profiles uploaded (and balanced) once, every pair through the option kernels
        values = _native.context().profile_distance_matrix([p.counts for p in profiles], profiles[0].length, options)
    else:
        values = [dist.distance(profiles[i], profiles[j]) for i in range(1, count) for j in range(i)]

    _write_matrix(output, count, values, precision)


def _write_matrix(output, count, values, precision):
    fmt = '{{0:.{0}f}}'.format(precision)
    at = 0
    for i in range(1, count):
        output.write(' '.join(fmt.format(values[at + j]) for j in range(i)))
        output.write('\n')
        at += i


def _device_pair(left, right):
    """(context, device address of left, of right) when both profiles' tables are still in HBM on one context, else None."""
    a = getattr(left, '_device_counts', None)
    b = getattr(right, '_device_counts', None)
    if a is None or b is None or left.length != right.length:
        return None
    a, b = a(), b()
    if not a or not b or a[0] is not b[0]:
        return None
    return a[0], a[1], b[1]


def _device_matrix(profiles, dist, metric):
    """The matrix values of profiles whose tables are ALL still in HBM (one context, one k) without a transfer: consecutive
    tables of one batch are used where they lie, anything else is gathered by device-to-device copies first.  None when a
    profile has host counts, or when a user-supplied callable keeps the distance in Python."""
    devs = []
    for p in profiles:
        d = getattr(p, '_device_counts', None)
        d = d() if d is not None else None
        if not d or (devs and d[0] is not devs[0][0]):
            return None
        devs.append(d)
    plain = dist._is_plain() and metric is not None and metric != _native.COSINE
    options = None if plain else dist._native_options()
    if not plain and options is None:
        return None
    ctx, k, P = devs[0][0], profiles[0].length, len(profiles)
    table_bytes = 8 * 4 ** k
    if not plain:        # the option kernels work pair by pair (balancing, positive, smoothing and scaling depend on the partner)
        return [ctx.profile_distance_device(k, devs[i][1], devs[j][1], options) for i in range(1, P) for j in range(i)]
    base, gathered = devs[0][1], None
    if any(devs[i][1] != base + i * table_bytes for i in range(P)):
        gathered = base = ctx.alloc(P * table_bytes)
        for i in range(P):
            ctx.d2d(base + i * table_bytes, devs[i][1], table_bytes)
    try:
        return ctx.distance_matrix_device(P, k, base, metric, do_balance=dist._do_balance)
    finally:
        if gathered is not None:
            ctx.sync()
            ctx.free(gathered)
