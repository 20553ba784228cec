#!/usr/bin/env python
"""End-to-end timing of the drop-in entry points on a multi-GB FASTA FILE: what a user of `kpal count`
gets, input on disk / in the page cache, not in HBM.

    python tools/clibench.py [--gb 8] [--k 12] [--dir /dev/shm] [--keep]

Writes a synthetic FASTA of --gb GB in 60-column lines (records of ~100 Mbases, bases from the SURVEY 8d generator on the
device), then times, each from a page-cached file:
  pread        the library's reader alone -- N threads pread into a buffer, no GPU (KPAL_READ_THREADS sweep): the host's ceiling
  feed_file    Context.count_feed_fasta_file + count_finish (read, H2D, flatten, count; table download)
  from_fasta   klib.Profile.from_fasta(open(path), k)          (the drop-in API on a text handle)
  cli_count    kpal_amd.kmer.main(['count', '-k', K, path, out])  (argparse, FileType handles, Profile.save; HDF5 through the in-memory
               stand-in of the tests -- h5py is not in this image)
  shards       the file cut for 8 ranks by kpal_amd.dist.fasta_shards, every shard through count_feed_fasta_file on this one GPU
               (cut cost + per-shard rate; the tables add up to the whole-file table: checked)
Prints one JSON object (bases/s are sequence bases, the file holds 61/60 bytes per base + headers)."""
import argparse
import ctypes
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np


def write_fasta(ctx, path, gb, record_bases=100_020_000):
    """-> (bytes, bases, records)."""
    width = 60
    lines_total = int(gb * 1e9) // (width + 1)
    per_record = record_bases // width
    piece_lines = 16_000_000                       # ~1 GB per device buffer
    d = ctx.alloc(piece_lines * (width + 1))
    host = np.empty(piece_lines * (width + 1), dtype=np.uint8)
    done, records, in_record = 0, 0, per_record
    with open(path, 'wb') as fh:
        while done < lines_total:
            n = min(piece_lines, lines_total - done)
            ctx.synth_reads_device(77, done, n, width, d)
            ctx.d2h(host[:n * (width + 1)], d)
            at = 0
            while at < n:
                if in_record == per_record:
                    records += 1
                    fh.write(b'>chr%d synthetic record\n' % records)
                    in_record = 0
                take = min(n - at, per_record - in_record)
                fh.write(host[at * (width + 1):(at + take) * (width + 1)].data)
                at += take
                in_record += take
            done += n
    ctx.free(d)
    return os.path.getsize(path), lines_total * width, records


def by_record_bench(ctx, args):
    """`--by-record`: Profile.from_fasta_by_record (records found on the device, names read from the header lines only) on files of
    many records: Gbases/s of the generator consumed to the end (every profile built, its total summed, then dropped), next to the
    host tokeniser alone (klib._fasta_records: the per-line interpreter loop the device index replaced) and to from_fasta on the
    same file.  The tables are n_records x 4^k x 8 bytes -- at k = 8 that is 52 x the input for 10 kb records: they, not the
    sequence, are what moves over PCIe and through NumPy."""
    from kpal_amd import klib
    shapes = [(100_000, 10_020, 4), (20_000, 10_020, 8), (1_000, 1_000_020, 8), (200_000, 300, 6)]
    if args.records:
        shapes = [(args.records, args.record_bases, args.k)]
    out = []
    for records, record_bases, k in shapes:
        path = os.path.join(args.dir, 'kpal_clibench_rec_%d.fa' % os.getpid())
        try:
            gb = records * (record_bases // 60) * 61 / 1e9
            nbytes, bases, nrec = write_fasta(ctx, path, gb, record_bases=record_bases)
            row = {'records': nrec, 'record_bases': record_bases, 'k': k, 'file_bytes': nbytes, 'bases': bases, 'table_bytes_total': nrec * 8 * 4 ** k}

            def consume(check):
                total, n, names = 0, 0, []
                with open(path) as fh:
                    for p in klib.Profile.from_fasta_by_record(fh, k):
                        if check:              # (summing the tables is the bench's own work: the first pass checks, the timed ones only iterate)
                            total += int(p.counts.sum())
                        n += 1
                        if n <= 2 or n == nrec:
                            names.append(p.name)
                return total, n, names
            total, n, names = consume(True)
            times = []
            for _ in range(args.repeat):
                t = time.perf_counter()
                _, n2, _ = consume(False)
                times.append(time.perf_counter() - t)
                assert n2 == n
            with open(path) as fh:
                whole = klib.Profile.from_fasta(fh, k)
            per_record_kmers = max(record_bases // 60 * 60 - k + 1, 0)
            assert n == nrec and names[:2] == ['chr1', 'chr2'] and names[-1] == 'chr%d' % nrec, (n, names)
            assert total <= int(whole.total) and total >= (nrec - 1) * per_record_kmers, (total, int(whole.total))
            row['by_record'] = {'s': min(times), 'Gbases_per_s': bases / min(times) / 1e9, 'records_per_s': nrec / min(times),
                                'tables_GBs': nrec * 8 * 4 ** k / min(times) / 1e9,
                                'note': 'the profiles are built and dropped; since round 6 their tables stay in HBM until something asks for counts'}
            # ... and with every table brought to the host (what `kpal count --by-record` needs for the HDF5 file)
            t = time.perf_counter()
            with open(path) as fh:
                got = sum(int(p.counts[0]) >= 0 for p in klib.Profile.from_fasta_by_record(fh, k))
            dt = time.perf_counter() - t
            assert got == nrec
            row['by_record_downloaded'] = {'s': dt, 'Gbases_per_s': bases / dt / 1e9, 'tables_GBs': nrec * 8 * 4 ** k / dt / 1e9}
            # the library flow the device tables are for: the records' profiles straight into a distance matrix (kmer.py:137-146 ->
            # 683-700 in one process), first 64 records
            import io
            from kpal_amd import kdistlib
            t = time.perf_counter()
            with open(path) as fh:
                gen = klib.Profile.from_fasta_by_record(fh, k)
                some = [p for _, p in zip(range(64), gen)]
                gen.close()
            out_text = io.StringIO()
            kdistlib.distance_matrix(some, out_text, 3, kdistlib.ProfileDistance())
            dt = time.perf_counter() - t
            row['first_64_records_to_matrix'] = {'s': dt, 'on_device': all(p._device_counts() is not None for p in some), 'lines': out_text.getvalue().count('\n')}
            del some
            t = time.perf_counter()
            with open(path) as fh:
                m = sum(len(sq) for _, sq in klib._fasta_records(fh))
            dt = time.perf_counter() - t
            assert m == bases
            row['host_tokeniser_alone'] = {'s': dt, 'Gbases_per_s': bases / dt / 1e9}
            out.append(row)
        finally:
            if os.path.exists(path):
                os.unlink(path)
    return out


def pread_rate(path, threads, limit=4 << 30):
    size = min(os.path.getsize(path), limit)
    buf = np.empty(64 << 20, dtype=np.uint8)
    fd = os.open(path, os.O_RDONLY)
    libc = ctypes.CDLL(None, use_errno=True)
    libc.pread.restype = ctypes.c_ssize_t
    libc.pread.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64]
    t0 = time.perf_counter()
    off = 0
    while off < size:
        n = min(buf.size, size - off)
        part = (n // threads + 4095) & ~4095

        def work(i):
            o = i * part
            if o < n:
                libc.pread(fd, buf.ctypes.data + o, min(part, n - o), off + o)
        ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        off += n
    dt = time.perf_counter() - t0
    os.close(fd)
    return size / dt / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gb', type=float, default=8.0)
    ap.add_argument('--k', type=int, default=12)
    ap.add_argument('--dir', default='/dev/shm')
    ap.add_argument('--keep', action='store_true')
    ap.add_argument('--repeat', type=int, default=2)
    ap.add_argument('--by-record', action='store_true', help='time Profile.from_fasta_by_record on files of many records instead')
    ap.add_argument('--records', type=int, default=0, help='--by-record: one shape only: this many records ...')
    ap.add_argument('--record-bases', type=int, default=10_020, help='... of this many bases each (at --k)')
    args = ap.parse_args()
    from kpal_amd import _native, dist, files, klib, kmer
    import memh5
    ctx = _native.context()
    if args.by_record:
        print(json.dumps({'by_record': by_record_bench(ctx, args), 'host_cores': os.cpu_count()}))
        return
    path = os.path.join(args.dir, 'kpal_clibench_%d.fa' % os.getpid())
    out = {'k': args.k, 'dir': args.dir, 'host_cores': os.cpu_count(), 'read_threads': int(os.environ.get('KPAL_READ_THREADS', '16')), 'read_pin': os.environ.get('KPAL_READ_PIN', '0')}
    try:
        t0 = time.perf_counter()
        nbytes, bases, records = write_fasta(ctx, path, args.gb)
        out.update(file_bytes=nbytes, bases=bases, records=records, write_s=time.perf_counter() - t0)
        out['pread_GBs'] = {str(t): pread_rate(path, t) for t in (1, 4, 8, 16, 32)}

        def best(fn):
            times = []
            for _ in range(args.repeat):
                t = time.perf_counter()
                r = fn()
                times.append(time.perf_counter() - t)
            return min(times), r

        def feed_file():
            ctx.count_begin(args.k)
            ctx.count_feed_fasta_file(path)
            return ctx.count_finish()
        s, table = best(feed_file)
        want_total = int(table.sum())
        out['feed_file'] = {'s': s, 'Gbases_per_s': bases / s / 1e9, 'file_GBs': nbytes / s / 1e9}

        def from_fasta():
            with open(path) as fh:
                profile = klib.Profile.from_fasta(fh, args.k)
                profile.counts          # (since round 6 the table stays in HBM until asked for: the download belongs to this figure)
                return profile
        s, p = best(from_fasta)
        assert int(p.total) == want_total
        out['from_fasta'] = {'s': s, 'Gbases_per_s': bases / s / 1e9, 'file_GBs': nbytes / s / 1e9}

        store = memh5.Store()
        files.open_profile_file = store.open

        import tempfile
        scratch = tempfile.mkdtemp(prefix='kpal_clibench_')

        def cli():
            name = os.path.join(scratch, 'out_%d.k%d' % (time.perf_counter_ns(), args.k))
            kmer.main(['count', '-k', str(args.k), path, name])
            return name
        s, name = best(cli)
        got = klib.Profile.from_file(store.open(name, 'r'))
        assert int(got.total) == want_total and np.array_equal(got.counts, table)
        out['cli_count'] = {'s': s, 'Gbases_per_s': bases / s / 1e9, 'file_GBs': nbytes / s / 1e9,
                            'note': 'kmer.main([count -k K file out]) in-process: argparse + FileType + from_fasta + Profile.save (in-memory HDF5 stand-in)'}

        t = time.perf_counter()
        shards = dist.fasta_shards(path, 8, args.k)
        cut_s = time.perf_counter() - t
        acc = np.zeros_like(table)
        per = []
        for segs in shards:
            t = time.perf_counter()
            dist.count_fasta_sharded(ctx, args.k, segs)
            c = ctx.count_finish()
            per.append(time.perf_counter() - t)
            acc += c
        assert np.array_equal(acc, table)
        out['shards'] = {'world': 8, 'cut_s': cut_s, 'per_shard_s': per, 'prefixed_segments': sum(1 for segs in shards for g in segs if g.prefix),
                         'sum_equals_whole_file': True, 'shard_Gbases_per_s': bases / 8 / max(per) / 1e9}
    finally:
        if not args.keep and os.path.exists(path):
            os.unlink(path)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
