"""ctypes binding of libkpal_hip.so (C-ABI: include/kpal_hip.h).

This is the only gateway from the Python host code to the HIP kernels.  There is NO CPU
fallback: if the shared library has not been built, or no GPU is visible, every hot-path call
raises -- loudly -- instead of computing something on the host.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('KPAL_HIP_LIBRARY', os.path.join(_HERE, 'libkpal_hip.so'))   # override: A/B timing of builds

KPAL_MAX_K = 16
PAIRWISE_PROD, PAIRWISE_SUM, EUCLIDEAN, COSINE = 0, 1, 2, 3
SUMMARY_MIN, SUMMARY_AVERAGE, SUMMARY_MEDIAN = 0, 1, 2
STRATEGY = {'auto': 0, 'global_atomic': 1, 'lds_direct': 2, 'partition': 3, 'partition2': 4, 'partition_chunked': 5, 'partition_quads': 6, 'partition2_quads': 7}

_E_INVALID, _E_NOMEM, _E_HIP, _E_STATE, _E_IO = -1, -2, -3, -4, -5

_lib = None
_lib_lock = threading.Lock()

_vp = ctypes.c_void_p
_i64p = ctypes.POINTER(ctypes.c_int64)
_f64p = ctypes.POINTER(ctypes.c_double)

class DistanceOptions(ctypes.Structure):
    """``kpal_distance_options`` of include/kpal_hip.h: the ProfileDistance constructor arguments
    (kpal/kdistlib.py:25-51) that select built-in behaviour."""
    _fields_ = [('do_balance', ctypes.c_int), ('do_positive', ctypes.c_int), ('do_smooth', ctypes.c_int),
                ('summary', ctypes.c_int), ('threshold', ctypes.c_double), ('do_scale', ctypes.c_int),
                ('down', ctypes.c_int), ('metric', ctypes.c_int)]


_optp = ctypes.POINTER(DistanceOptions)


class ProfileStats(ctypes.Structure):
    """``kpal_profile_stats`` of include/kpal_hip.h (kpal/klib.py:193-225)."""
    _fields_ = [('total', ctypes.c_int64), ('non_zero', ctypes.c_int64), ('min', ctypes.c_int64),
                ('max', ctypes.c_int64), ('mean', ctypes.c_double), ('median', ctypes.c_double),
                ('std', ctypes.c_double)]


_statp = ctypes.POINTER(ProfileStats)
MERGE_SUM, MERGE_XOR, MERGE_INT, MERGE_NINT = 0, 1, 2, 3

# name -> (restype, argtypes); one entry per symbol declared in include/kpal_hip.h
SIGNATURES = {
    'kpal_last_error': (ctypes.c_char_p, []),
    'kpal_version': (ctypes.c_char_p, []),
    'kpal_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'kpal_ctx_create': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp)]),
    'kpal_ctx_destroy': (None, [_vp]),
    'kpal_sync': (ctypes.c_int, [_vp]),
    'kpal_dev_alloc': (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    'kpal_dev_free': (ctypes.c_int, [_vp, _vp]),
    'kpal_memcpy_h2d': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    'kpal_memcpy_d2h': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    'kpal_memcpy_d2d': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    'kpal_count_begin': (ctypes.c_int, [_vp, ctypes.c_int]),
    'kpal_count_set_strategy': (ctypes.c_int, [_vp, ctypes.c_int]),
    'kpal_count_feed': (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    'kpal_count_feed_device': (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    'kpal_count_feed_pinned': (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    'kpal_host_alloc': (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    'kpal_host_free': (ctypes.c_int, [_vp, _vp]),
    'kpal_count_feed_fasta': (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    'kpal_count_feed_fasta_file': (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, _vp, ctypes.c_size_t]),
    'kpal_fasta_flatten': (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp, ctypes.POINTER(ctypes.c_uint64)]),
    'kpal_count_records': (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_size_t, _vp, ctypes.c_size_t, _vp]),
    'kpal_fasta_records_begin': (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    'kpal_fasta_records_index': (ctypes.c_int, [_vp, _vp, _vp]),
    'kpal_fasta_records_count_device': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    'kpal_fasta_records_file_open': (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64]),
    'kpal_fasta_records_file_next': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int)]),
    'kpal_fasta_records_file_tell': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64)]),
    'kpal_fasta_records_file_close': (ctypes.c_int, [_vp]),
    'kpal_fasta_records_count': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    'kpal_count_finish': (ctypes.c_int, [_vp, _vp]),
    'kpal_count_table': (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_uint64)]),
    'kpal_count_balance': (ctypes.c_int, [_vp]),
    'kpal_count_last_plan': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    'kpal_count_stats': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]),
    'kpal_comm_probe': (ctypes.c_int, [ctypes.c_char_p]),
    'kpal_comm_unique_id': (ctypes.c_int, [ctypes.c_char_p, _vp]),
    'kpal_comm_init': (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, _vp]),
    'kpal_comm_destroy': (ctypes.c_int, [_vp]),
    'kpal_comm_reduce_table': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    'kpal_comm_reduce_table_async': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    'kpal_comm_merged_table': (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_uint64)]),
    'kpal_comm_max_f64': (ctypes.c_int, [_vp, _f64p]),
    'kpal_comm_reduce_scatter_table': (ctypes.c_int, [_vp, ctypes.c_int]),
    'kpal_comm_gather_table': (ctypes.c_int, [_vp]),
    'kpal_comm_merged_range': (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    'kpal_range_pack_device': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'kpal_range_unpack_device': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'kpal_comm_distance_matrix_device': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_uint64, _vp, ctypes.c_int, _f64p]),
    'kpal_synth_reads_device': (ctypes.c_int, [_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64,
                                               ctypes.c_int, ctypes.c_int, _vp]),
    'kpal_balance': (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    'kpal_balance_device': (ctypes.c_int, [_vp, ctypes.c_int, _vp]),
    'kpal_reverse_complement': (ctypes.c_uint64, [ctypes.c_uint64, ctypes.c_int]),
    'kpal_split': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint64)]),
    'kpal_strand_balance': (ctypes.c_int, [_vp, ctypes.c_int, _vp, ctypes.c_int, _f64p]),
    'kpal_pair_distance': (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, _f64p, _i64p]),
    'kpal_pair_distance_device': (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp, ctypes.c_int, ctypes.c_int,
                                                 ctypes.c_int, _f64p, _i64p]),
    'kpal_pair_distance_f64': (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp, ctypes.c_int, _f64p, _i64p]),
    'kpal_distance_matrix': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp), ctypes.c_int,
                                            ctypes.c_int, _f64p]),
    'kpal_distance_matrix_device': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, ctypes.c_int,
                                                   ctypes.c_int, _f64p]),
    'kpal_profile_distance': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _optp, _f64p]),
    'kpal_profile_distance_device': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, _optp, _f64p]),
    'kpal_dynamic_smooth': (ctypes.c_int, [_vp, ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_double]),
    'kpal_profile_distance_matrix': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp), _optp,
                                                    _f64p]),
    'kpal_stats': (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _statp]),
    'kpal_stats_device': (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _statp]),
    'kpal_merge': (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp, ctypes.c_int, _vp]),
    'kpal_merge_device': (ctypes.c_int, [_vp, ctypes.c_size_t, _vp, _vp, ctypes.c_int, _vp]),
    'kpal_shrink': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'kpal_shrink_device': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'kpal_prof_enable': (ctypes.c_int, [_vp, ctypes.c_int]),
    'kpal_prof_reset': (ctypes.c_int, [_vp]),
    'kpal_prof_count': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int)]),
    'kpal_prof_get': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, _f64p,
                                     ctypes.POINTER(ctypes.c_uint64)]),
}


class NativeLibraryMissing(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so /
    libhsa-runtime64.so; if libkpal_hip.so pulled in /opt/rocm's copy first, a later ``import torch``
    would start a second runtime that finds no GPU.  When PyTorch is installed (it is only needed
    for the multi-GPU reduce), load its copies first -- libkpal_hip.so then binds to them by soname,
    whichever of the two packages the caller imports first.  Does not import torch."""
    import sys
    if 'torch' in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], 'lib')
    for name in ('libhsa-runtime64.so', 'libamdhip64.so'):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return


def load():
    """Load libkpal_hip.so (once).  Raises NativeLibraryMissing if it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NativeLibraryMissing(
                    '%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                    '(hipcc --offload-arch=gfx950).  kpal_amd has no CPU fallback.' % LIB_PATH)
            _share_hip_runtime_with_torch()
            L = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                if 'KPAL_HIP_LIBRARY' in os.environ and not hasattr(L, name):
                    continue   # A/B timing against an older build that lacks newer entry points
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


def _check(rc):
    if rc == 0:
        return
    msg = load().kpal_last_error().decode('utf-8', 'replace')
    if rc == _E_INVALID:
        raise ValueError(msg)
    if rc == _E_NOMEM:
        raise MemoryError(msg)
    if rc == _E_IO:
        raise OSError(msg)
    raise RuntimeError('kpal_hip error %d: %s' % (rc, msg))


COMM_ID_BYTES = 128


def rccl_library():
    """Path of the librccl.so the library should bind (bytes), or None for its own search: inside a process that has
    PyTorch installed its bundled copy is taken -- the one that matches the HIP runtime ``_share_hip_runtime_with_torch``
    loaded; KPAL_RCCL_LIBRARY overrides."""
    env = os.environ.get('KPAL_RCCL_LIBRARY')
    if env:
        return env.encode()
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.submodule_search_locations:
        path = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'librccl.so')
        if os.path.exists(path):
            return path.encode()
    return None


def comm_unique_id():
    """128 opaque bytes (an ncclUniqueId) that rank 0 creates and every rank passes to ``Context.comm_init``."""
    buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
    _check(load().kpal_comm_unique_id(rccl_library(), buf))
    return bytes(buf)


def comm_probe():
    """Can libkpal_hip.so bind RCCL in this process (dlopen + symbols)?  Raises if not.  What ranks > 0 call instead of
    ``comm_unique_id`` before the collective ``Context.comm_init``."""
    _check(load().kpal_comm_probe(rccl_library()))


def device_count():
    n = ctypes.c_int(0)
    rc = load().kpal_device_count(ctypes.byref(n))
    if rc != 0:
        return 0
    return n.value


def reverse_complement(number, length):
    return int(load().kpal_reverse_complement(int(number) & 0xFFFFFFFFFFFFFFFF, int(length)))


def _as_i64(a, what='counts'):
    a = np.asanyarray(a)
    if a.dtype.kind not in 'iub':
        raise TypeError('%s must be an integer array (got %s)' % (what, a.dtype))
    return np.ascontiguousarray(a, dtype=np.int64)


class DeviceTable(object):
    """Zero-copy view of a device buffer through __cuda_array_interface__ (consumed by
    torch.as_tensor(..., device='cuda') for the RCCL reduce)."""

    def __init__(self, ptr, n, owner, typestr='<i8'):
        self.ptr, self.n, self._owner = ptr, n, owner
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': typestr, 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


class Context(object):
    """One GPU, one HIP stream.  Not thread-safe (mirrors the single-threaded reference)."""

    def __init__(self, device=0):
        self._L = load()
        h = _vp()
        _check(self._L.kpal_ctx_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, '_h', None):
            self._gather_buffer = None             # (klib: the page-locked gather buffer of from_sequences -- freed with the context)
            self._L.kpal_ctx_destroy(self._h)      # (also releases what host_alloc handed out: views of it dangle from here on)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- memory ------------------------------------------------------------------------------
    def alloc(self, nbytes):
        p = _vp()
        _check(self._L.kpal_dev_alloc(self._h, int(nbytes), ctypes.byref(p)))
        return p.value

    def free(self, ptr):
        _check(self._L.kpal_dev_free(self._h, _vp(ptr)))

    def h2d(self, dev_ptr, host_array):
        a = np.ascontiguousarray(host_array)
        _check(self._L.kpal_memcpy_h2d(self._h, _vp(dev_ptr), a.ctypes.data, a.nbytes))

    def d2h(self, host_array, dev_ptr):
        assert host_array.flags['C_CONTIGUOUS']
        _check(self._L.kpal_memcpy_d2h(self._h, host_array.ctypes.data, _vp(dev_ptr), host_array.nbytes))

    def d2d(self, dev_dst, dev_src, nbytes):
        _check(self._L.kpal_memcpy_d2d(self._h, _vp(dev_dst), _vp(dev_src), int(nbytes)))

    def sync(self):
        _check(self._L.kpal_sync(self._h))

    # -- counting ----------------------------------------------------------------------------
    def count_begin(self, k, strategy='auto'):
        _check(self._L.kpal_count_set_strategy(self._h, STRATEGY[strategy]))
        _check(self._L.kpal_count_begin(self._h, int(k)))

    def count_feed(self, buf):
        """buf: bytes-like or uint8 array in host memory."""
        a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf, dtype=np.uint8)
        if a.size:
            _check(self._L.kpal_count_feed(self._h, a.ctypes.data, a.size))

    def host_alloc(self, nbytes):
        """Page-locked host buffer -> (address, uint8 NumPy view); free with ``host_free(address)`` (the view dangles afterwards)."""
        p = _vp()
        _check(self._L.kpal_host_alloc(self._h, int(nbytes), ctypes.byref(p)))
        view = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(int(nbytes),))
        return p.value, view

    def host_free(self, address):
        _check(self._L.kpal_host_free(self._h, _vp(address)))

    def count_feed_pinned(self, address, nbytes):
        """``nbytes`` at ``address`` inside a ``host_alloc`` buffer: copied by DMA in place; returns when the buffer may be refilled."""
        if nbytes:
            _check(self._L.kpal_count_feed_pinned(self._h, _vp(address), int(nbytes)))

    def count_feed_fasta(self, buf):
        """buf: FASTA text (bytes-like) made of whole records; flattened and counted on the GPU."""
        a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf, dtype=np.uint8)
        if a.size:
            _check(self._L.kpal_count_feed_fasta(self._h, a.ctypes.data, a.size))

    def count_feed_fasta_file(self, path, begin=0, end=0, prefix=b''):
        """The bytes [begin, end) of a FASTA file (end = 0: to its end), read by the library itself (parallel preads into its
        pinned staging buffers), flattened and counted on the GPU, chunks pipelined.  ``prefix``: text that logically precedes
        the range (a shard that begins inside a record: ``b'>\\n'`` + the k - 1 bases before it)."""
        prefix = bytes(prefix)
        pbuf = (ctypes.c_uint8 * max(len(prefix), 1)).from_buffer_copy(prefix or b'\0')
        _check(self._L.kpal_count_feed_fasta_file(self._h, os.fsencode(path), int(begin), int(end), pbuf, len(prefix)))

    def fasta_flatten(self, buf):
        """-> the flat byte stream the counting kernels see for this FASTA text (tests)."""
        a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf, dtype=np.uint8)
        out = np.empty(max(a.size, 1), dtype=np.uint8)
        n = ctypes.c_uint64(0)
        _check(self._L.kpal_fasta_flatten(self._h, a.ctypes.data if a.size else None, a.size, out.ctypes.data, ctypes.byref(n)))
        return out[:n.value].tobytes()

    def count_feed_device(self, dev_ptr, nbytes):
        _check(self._L.kpal_count_feed_device(self._h, _vp(dev_ptr), int(nbytes)))

    def count_finish(self, k=None, to_host=True):
        if not to_host:
            _check(self._L.kpal_count_finish(self._h, None))
            return None
        ptr, n = self.count_table()
        out = np.empty(n, dtype=np.int64)
        _check(self._L.kpal_count_finish(self._h, out.ctypes.data))
        return out

    def count_table(self):
        p = _vp()
        n = ctypes.c_uint64(0)
        _check(self._L.kpal_count_table(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def count_balance(self):
        """Profile.balance on the device count table, in place (fused into the finalisation of the table for k >= 13)."""
        _check(self._L.kpal_count_balance(self._h))

    def count_last_plan(self):
        """-> (strategy name, steps1, steps2): the pipeline and quad tile sizes the last piece of the last feed took."""
        st, s1, s2 = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        _check(self._L.kpal_count_last_plan(self._h, ctypes.byref(st), ctypes.byref(s1), ctypes.byref(s2)))
        names = {v: n for n, v in STRATEGY.items()}
        return names.get(st.value, str(st.value)), s1.value, s2.value

    def count_stats(self):
        """Cumulative slow-path statistics of the quad pipelines since the context was created (kpal_count_stats)."""
        out = (ctypes.c_uint64 * 9)()
        _check(self._L.kpal_count_stats(self._h, out, 9))
        names = ('hot_entries', 'spilled_items', 'unlisted_items', 'fresh_pieces', 'fresh_reruns', 'quad_pieces', 'chunked_pieces', 'split_pieces', 'repeat_pieces')
        return dict(zip(names, (int(v) for v in out)))

    def count_table_view(self):
        ptr, n = self.count_table()
        return DeviceTable(ptr, n, self)

    def synth_reads_device(self, seed, first_read, n_reads, read_len, dev_ptr, noisy=False):
        _check(self._L.kpal_synth_reads_device(self._h, int(seed), int(first_read), int(n_reads), int(read_len),
                                               int(bool(noisy)), _vp(dev_ptr)))

    def count_records(self, k, flat, starts):
        """One table per record of a flat byte stream: record r = flat[starts[r]:starts[r+1]]
        (separator bytes between records) -> int64[n_records, 4**k]."""
        a = np.frombuffer(flat, dtype=np.uint8) if not isinstance(flat, np.ndarray) else np.ascontiguousarray(flat, dtype=np.uint8)
        st = np.ascontiguousarray(starts, dtype=np.uint64)
        n = st.size - 1
        out = np.empty((max(n, 0), 4 ** k), dtype=np.int64)
        if n > 0:
            _check(self._L.kpal_count_records(self._h, int(k), a.ctypes.data if a.size else None, a.size, st.ctypes.data, n,
                                              out.ctypes.data))
        return out

    def fasta_records_begin(self, text):
        """Tokenise FASTA text (whole records) on the device -> (n_records, flattened bytes); the index stays in the context."""
        a = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else np.ascontiguousarray(text, dtype=np.uint8)
        n, nf = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(self._L.kpal_fasta_records_begin(self._h, a.ctypes.data if a.size else None, a.size, ctypes.byref(n), ctypes.byref(nf)))
        self._records = n.value
        self._records_scan = getattr(self, '_records_scan', 0) + 1      # (a scan interleaved with another one would read the other's index)
        return n.value, nf.value

    def fasta_records_file_open(self, path, begin=0, end=0):
        """Start a by-record scan of bytes [begin, end) of a FASTA file the library reads itself (end = 0: to its end)."""
        _check(self._L.kpal_fasta_records_file_open(self._h, os.fsencode(path), int(begin), int(end)))

    def fasta_records_file_next(self):
        """Index the next piece of whole records -> (n_records, flattened bytes, file offset of the piece), or None at the end."""
        n, nf, off, done = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_int(0)
        _check(self._L.kpal_fasta_records_file_next(self._h, ctypes.byref(n), ctypes.byref(nf), ctypes.byref(off), ctypes.byref(done)))
        self._records = n.value
        self._records_scan = getattr(self, '_records_scan', 0) + 1
        return None if done.value else (n.value, nf.value, off.value)

    def fasta_records_file_tell(self):
        """File offset of the first byte that no piece has covered yet (where a closed scan is opened again)."""
        off = ctypes.c_uint64(0)
        _check(self._L.kpal_fasta_records_file_tell(self._h, ctypes.byref(off)))
        return off.value

    def fasta_records_file_close(self):
        _check(self._L.kpal_fasta_records_file_close(self._h))

    def fasta_records_index(self):
        """-> (offset of every record's header line in the text, start of every record in the flattened stream [n + 1])."""
        n = getattr(self, '_records', 0)
        hdr = np.empty(n, dtype=np.uint64)
        starts = np.empty(n + 1, dtype=np.uint64)
        if n:
            _check(self._L.kpal_fasta_records_index(self._h, hdr.ctypes.data, starts.ctypes.data))
        return hdr, starts

    def fasta_records_count(self, k, first, n):
        """Tables of records [first, first + n) of the indexed text -> int64[n, 4**k]."""
        out = np.empty((max(n, 0), 4 ** k), dtype=np.int64)
        if n > 0:
            _check(self._L.kpal_fasta_records_count(self._h, int(k), int(first), int(n), out.ctypes.data))
        return out

    def fasta_records_count_device(self, k, first, n, dev_out):
        """Tables of records [first, first + n) of the indexed text into n * 4**k int64 of device memory at ``dev_out``."""
        if n > 0:
            _check(self._L.kpal_fasta_records_count_device(self._h, int(k), int(first), int(n), _vp(dev_out)))

    def count_bytes(self, k, buf, strategy='auto'):
        """Count one flat host byte stream -> int64[4**k]."""
        self.count_begin(k, strategy)
        self.count_feed(buf)
        return self.count_finish()

    # -- multi-GPU (RCCL inside the library) -----------------------------------------------------------
    def comm_init(self, rank, world, comm_id):
        """Join the RCCL communicator identified by ``comm_id`` (``comm_unique_id()`` of rank 0)."""
        buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        _check(self._L.kpal_comm_init(self._h, rccl_library(), int(rank), int(world), buf))

    def comm_destroy(self):
        _check(self._L.kpal_comm_destroy(self._h))

    def comm_reduce_table(self, root=0, balance=True, pipelined=False):
        """ONE ncclReduce(int64, sum) of the count tables onto ``root`` (+ Profile.balance there), queued on the
        context's streams.  ``pipelined``: on a copy of the table and a second stream -- the next count overlaps it."""
        fn = self._L.kpal_comm_reduce_table_async if pipelined else self._L.kpal_comm_reduce_table
        _check(fn(self._h, int(root), int(bool(balance))))

    def comm_merged_table(self):
        p = _vp()
        n = ctypes.c_uint64(0)
        _check(self._L.kpal_comm_merged_table(self._h, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    def comm_reduce_scatter_table(self, balance=False):
        """The bin-range merge: ONE ncclReduceScatter leaves this rank its range of the merged table (in place), balanced through
        one all-to-all of the mirrored entries when asked (power-of-two worlds)."""
        _check(self._L.kpal_comm_reduce_scatter_table(self._h, int(bool(balance))))

    def comm_gather_table(self):
        """After comm_reduce_scatter_table: every rank's table completed by one ncclAllGather (collective)."""
        _check(self._L.kpal_comm_gather_table(self._h))

    def comm_merged_range(self):
        """-> (device pointer, first bin, number of bins) of this rank's part of the merged table."""
        p, first, n = _vp(), ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(self._L.kpal_comm_merged_range(self._h, ctypes.byref(p), ctypes.byref(first), ctypes.byref(n)))
        return p.value, first.value, n.value

    def range_pack_device(self, k, rank, world, dev_table, dev_send):
        _check(self._L.kpal_range_pack_device(self._h, int(k), int(rank), int(world), _vp(dev_table), _vp(dev_send)))

    def range_unpack_device(self, k, rank, world, dev_table, dev_recv):
        _check(self._L.kpal_range_unpack_device(self._h, int(k), int(rank), int(world), _vp(dev_table), _vp(dev_recv)))

    def comm_distance_matrix_device(self, P, bin_count, dev_slices, metric):
        """distance_matrix values from bin-range shards: this rank's int64[P][bin_count] slices on the device -> the full lower
        triangle on every rank (one all-reduce of the per-pair partial sums and counts)."""
        out = np.zeros(P * (P - 1) // 2, dtype=np.float64)
        _check(self._L.kpal_comm_distance_matrix_device(self._h, int(P), int(bin_count), _vp(dev_slices), int(metric),
                                                        out.ctypes.data_as(_f64p)))
        return out

    def comm_max(self, value):
        v = ctypes.c_double(float(value))
        _check(self._L.kpal_comm_max_f64(self._h, ctypes.byref(v)))
        return v.value

    # -- vector operations -----------------------------------------------------------------------
    def balance_inplace(self, counts, k):
        assert counts.dtype == np.int64 and counts.flags['C_CONTIGUOUS'] and counts.flags['WRITEABLE']
        _check(self._L.kpal_balance(self._h, int(k), counts.ctypes.data))

    def balance_device(self, k, dev_ptr):
        _check(self._L.kpal_balance_device(self._h, int(k), _vp(dev_ptr)))

    def split(self, counts, k):
        c = _as_i64(counts)
        f = np.empty(c.size, dtype=np.int64)
        r = np.empty(c.size, dtype=np.int64)
        m = ctypes.c_uint64(0)
        _check(self._L.kpal_split(self._h, int(k), c.ctypes.data, f.ctypes.data, r.ctypes.data, ctypes.byref(m)))
        return f[:m.value].copy(), r[:m.value].copy()

    def strand_balance(self, counts, k, pairwise=PAIRWISE_PROD):
        c = _as_i64(counts)
        out = ctypes.c_double(0.0)
        _check(self._L.kpal_strand_balance(self._h, int(k), c.ctypes.data, int(pairwise), ctypes.byref(out)))
        return out.value

    def pair_distance(self, left, right, metric, do_balance=False, k=0, return_aux=False):
        l = _as_i64(left, 'left')
        r = _as_i64(right, 'right')
        if l.shape != r.shape or l.ndim != 1:
            raise ValueError('left and right must be 1-d vectors of equal length')
        out = ctypes.c_double(0.0)
        aux = ctypes.c_int64(0)
        _check(self._L.kpal_pair_distance(self._h, l.size, l.ctypes.data, r.ctypes.data, int(metric),
                                          int(bool(do_balance)), int(k), ctypes.byref(out), ctypes.byref(aux)))
        return (out.value, aux.value) if return_aux else out.value

    def pair_distance_f64(self, left, right, pairwise, return_aux=False):
        l = np.ascontiguousarray(left, dtype=np.float64)
        r = np.ascontiguousarray(right, dtype=np.float64)
        if l.shape != r.shape or l.ndim != 1:
            raise ValueError('left and right must be 1-d vectors of equal length')
        out = ctypes.c_double(0.0)
        aux = ctypes.c_int64(0)
        _check(self._L.kpal_pair_distance_f64(self._h, l.size, l.ctypes.data, r.ctypes.data, int(pairwise),
                                              ctypes.byref(out), ctypes.byref(aux)))
        return (out.value, aux.value) if return_aux else out.value

    def pair_distance_device(self, n, dev_left, dev_right, metric, do_balance=False, k=0):
        out = ctypes.c_double(0.0)
        aux = ctypes.c_int64(0)
        _check(self._L.kpal_pair_distance_device(self._h, int(n), _vp(dev_left), _vp(dev_right), int(metric),
                                                 int(bool(do_balance)), int(k), ctypes.byref(out), ctypes.byref(aux)))
        return out.value

    def distance_matrix(self, profiles, k, metric, do_balance=False):
        """profiles: list of int64[4**k] host vectors -> float64[P(P-1)/2] (kdistlib.py:179-186 order)."""
        arrs = [_as_i64(p) for p in profiles]
        P = len(arrs)
        for a in arrs:
            if a.size != 4 ** k:
                raise ValueError('profile length %d != 4**%d' % (a.size, k))
        out = np.zeros(P * (P - 1) // 2, dtype=np.float64)
        ptrs = (_vp * P)(*[a.ctypes.data for a in arrs])
        _check(self._L.kpal_distance_matrix(self._h, P, int(k), ptrs, int(metric), int(bool(do_balance)),
                                            out.ctypes.data_as(_f64p)))
        return out

    def distance_matrix_device(self, P, k, dev_profiles, metric, do_balance=False):
        out = np.zeros(P * (P - 1) // 2, dtype=np.float64)
        _check(self._L.kpal_distance_matrix_device(self._h, int(P), int(k), _vp(dev_profiles), int(metric),
                                                   int(bool(do_balance)), out.ctypes.data_as(_f64p)))
        return out

    # -- ProfileDistance with options ------------------------------------------------------------
    def profile_distance(self, left, right, k, options):
        """ProfileDistance.distance for a DistanceOptions (kdistlib.py:126-161)."""
        l = _as_i64(left, 'left')
        r = _as_i64(right, 'right')
        if l.size != 4 ** k or r.size != 4 ** k:
            raise ValueError('profile length != 4**%d' % k)
        out = ctypes.c_double(0.0)
        _check(self._L.kpal_profile_distance(self._h, int(k), l.ctypes.data, r.ctypes.data, ctypes.byref(options),
                                             ctypes.byref(out)))
        return out.value

    def profile_distance_device(self, k, dev_left, dev_right, options):
        out = ctypes.c_double(0.0)
        _check(self._L.kpal_profile_distance_device(self._h, int(k), _vp(dev_left), _vp(dev_right),
                                                    ctypes.byref(options), ctypes.byref(out)))
        return out.value

    def dynamic_smooth(self, left, right, k, summary, threshold):
        """In place on two writable contiguous int64 vectors (kdistlib.py:112-124)."""
        _check(self._L.kpal_dynamic_smooth(self._h, int(k), left.ctypes.data, right.ctypes.data, int(summary),
                                           float(threshold)))

    def stats(self, counts):
        """total / non_zero / min / max / mean / median / std of an int64 vector (klib.py:193-225)."""
        c = _as_i64(counts)
        out = ProfileStats()
        _check(self._L.kpal_stats(self._h, c.size, c.ctypes.data, ctypes.byref(out)))
        return out

    def stats_device(self, dev_counts, n):
        out = ProfileStats()
        _check(self._L.kpal_stats_device(self._h, int(n), dev_counts, ctypes.byref(out)))
        return out

    def merge(self, left, right, merger):
        """metrics.mergers[...](left, right) on int64 vectors (metrics.py:174-179) -> new array."""
        l, r = _as_i64(left), _as_i64(right)
        if l.size != r.size:
            raise ValueError('vectors differ in length: %d != %d' % (l.size, r.size))
        out = np.empty(l.size, dtype=np.int64)
        _check(self._L.kpal_merge(self._h, l.size, l.ctypes.data, r.ctypes.data, int(merger), out.ctypes.data))
        return out

    def shrink(self, counts, k, factor):
        """Sums of 4**factor consecutive counts (klib.py:329-352) -> new array of 4**(k-factor)."""
        c = _as_i64(counts)
        if c.size != 4 ** k:
            raise ValueError('profile length %d != 4**%d' % (c.size, k))
        if not 1 <= factor < k:
            raise ValueError('Reduction factor should be smaller than k-mer size.')
        out = np.empty(4 ** (k - factor), dtype=np.int64)
        _check(self._L.kpal_shrink(self._h, int(k), int(factor), c.ctypes.data, out.ctypes.data))
        return out

    def profile_distance_matrix(self, profiles, k, options):
        arrs = [_as_i64(p) for p in profiles]
        P = len(arrs)
        for a in arrs:
            if a.size != 4 ** k:
                raise ValueError('profile length %d != 4**%d' % (a.size, k))
        out = np.zeros(P * (P - 1) // 2, dtype=np.float64)
        ptrs = (_vp * P)(*[a.ctypes.data for a in arrs])
        _check(self._L.kpal_profile_distance_matrix(self._h, P, int(k), ptrs, ctypes.byref(options),
                                                    out.ctypes.data_as(_f64p)))
        return out

    # -- profiling ---------------------------------------------------------------------------
    def prof_enable(self, on=True):
        _check(self._L.kpal_prof_enable(self._h, int(bool(on))))

    def prof_reset(self):
        _check(self._L.kpal_prof_reset(self._h))

    def prof_get(self):
        """-> {kernel name: (total_ms, launches)}"""
        n = ctypes.c_int(0)
        _check(self._L.kpal_prof_count(self._h, ctypes.byref(n)))
        out = {}
        for i in range(n.value):
            name = ctypes.create_string_buffer(64)
            ms = ctypes.c_double(0.0)
            cnt = ctypes.c_uint64(0)
            _check(self._L.kpal_prof_get(self._h, i, name, 64, ctypes.byref(ms), ctypes.byref(cnt)))
            out[name.value.decode()] = (ms.value, cnt.value)
        return out


_default_ctx = None


def default_device():
    """LOCAL_RANK when launched one-process-per-GPU by torch.distributed.run, else KPAL_DEVICE or 0."""
    return int(os.environ.get('KPAL_DEVICE', os.environ.get('LOCAL_RANK', '0')))


def context():
    """Process-wide default Context (created on first use; raises if no library / no GPU)."""
    global _default_ctx
    if _default_ctx is None:
        if device_count() < 1:
            raise RuntimeError('kpal_amd: no HIP device visible; the k-mer hot path has no CPU fallback')
        _default_ctx = Context(default_device())
    return _default_ctx
