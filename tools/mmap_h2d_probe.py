#!/usr/bin/env python
"""Probe (GPU box): how fast does a FASTA file's text reach HBM without passing through a staging copy?
hipMemcpy straight from the pages of an mmap'ed file (tmpfs and disk page cache) against the library's pread-into-pinned
pipeline.   python tools/mmap_h2d_probe.py [--gb 4] [--dir /dev/shm]"""
import argparse, ctypes, mmap, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kpal_amd import _native

ap = argparse.ArgumentParser()
ap.add_argument('--gb', type=float, default=4.0)
ap.add_argument('--dir', default='/dev/shm')
a = ap.parse_args()
ctx = _native.context()
n = int(a.gb * 1e9) // 4096 * 4096
path = os.path.join(a.dir, 'kpal_mmap_probe_%d.bin' % os.getpid())
buf = np.random.randint(65, 85, n, dtype=np.uint8)
buf.tofile(path)
d = ctx.alloc(n)
try:
    t = time.perf_counter(); ctx.h2d(d, buf); print('anonymous pageable array  %.1f GB/s' % (n / (time.perf_counter() - t) / 1e9))
    t = time.perf_counter(); ctx.h2d(d, buf); print('anonymous pageable array  %.1f GB/s (again)' % (n / (time.perf_counter() - t) / 1e9))
    for flags, name in ((mmap.MAP_SHARED, 'MAP_SHARED'), (mmap.MAP_PRIVATE, 'MAP_PRIVATE')):
        for populate in (0, getattr(mmap, 'MAP_POPULATE', 0)):
            fd = os.open(path, os.O_RDONLY)
            t0 = time.perf_counter()
            mm = mmap.mmap(fd, n, flags=flags | populate, prot=mmap.PROT_READ)
            tm = time.perf_counter() - t0
            addr = ctypes.addressof(ctypes.c_char.from_buffer_copy(b'x'))  # placeholder
            arr = np.frombuffer(mm, dtype=np.uint8)
            for rep in range(2):
                t = time.perf_counter()
                rc = ctx._L.kpal_memcpy_h2d(ctx._h, ctypes.c_void_p(d), ctypes.c_void_p(arr.ctypes.data), n)
                dt = time.perf_counter() - t
                print('%-11s populate=%d rep %d: rc=%d  %.1f GB/s  (mmap %.3f s)' % (name, 1 if populate else 0, rep, rc, n / dt / 1e9, tm))
            # in 64 MiB pieces, as a pipeline would issue them
            t = time.perf_counter()
            step = 64 << 20
            for off in range(0, n, step):
                ctx._L.kpal_memcpy_h2d(ctx._h, ctypes.c_void_p(d + off), ctypes.c_void_p(arr.ctypes.data + off), min(step, n - off))
            print('%-11s populate=%d 64 MiB pieces: %.1f GB/s' % (name, 1 if populate else 0, n / (time.perf_counter() - t) / 1e9))
            del arr
            mm.close()
            os.close(fd)
finally:
    os.unlink(path)
