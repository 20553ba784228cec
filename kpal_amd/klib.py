"""Drop-in for ``kpal.klib``: the :class:`Profile` container with k-mer counting, balance and
split running as HIP kernels on an MI355X.

Same constructor, classmethods, methods, properties and error behaviour as the reference class
(kpal/klib.py:26-487).  Differences are internal only:

* ``from_sequences`` / ``from_fasta`` do not loop over characters in Python: sequences are joined
  into a flat byte stream (one ``\\n`` between sequences -- any byte outside ``AaCcGgTt`` splits
  windows exactly like the reference's ``re.split``, kpal/klib.py:152-156) and fed to
  ``kpal_count_feed``.
* FASTA is tokenised here (the reference delegates to ``Bio.SeqIO.parse``, kpal/klib.py:111,131):
  lines starting with ``>`` open a record whose name is the first whitespace-delimited token;
  the record's remaining lines are concatenated with surrounding whitespace removed; text before
  the first ``>`` is ignored.
* ``balance`` / ``split`` call ``kpal_balance`` / ``kpal_split``.

There is no CPU fallback for those paths: without libkpal_hip.so or without a GPU they raise.
Container-only helpers (``merge``, ``shrink``, ``shuffle``, statistics, printers, HDF5 I/O) are
not part of the hot path and use NumPy like the reference.
"""
import codecs
import io
import itertools
import math
import os
import stat

import numpy as np

from . import _native, metrics

# The C gatherer of from_sequences (csrc/kpal_gather.c, built by __graft_entry__.build(): walks the list's objects and copies them on
# several threads); without it the sequences are joined by the interpreter -- same stream, ~3 x slower for lists of short reads.
try:
    from . import _kpal_gather
except ImportError:       # pragma: no cover
    _kpal_gather = None

_FEED_BYTES = 32 << 20  # host-side join buffer per feed
_GATHER_BYTES = 64 << 20  # page-locked gather buffer of from_sequences
_GATHER_THREADS = max(1, min(64, int(os.environ.get('KPAL_GATHER_THREADS', '16'))))   # threads of the gatherer (walk + copies; 4 / 8 / 16 / 32: 114 / 88 / 80 / 92 ms for 8 M reads)
_JOIN_BLOCK = 4096       # sequences joined per C-level call
_RECORD_BATCH_BYTES = 1 << 30   # tables counted per from_fasta_by_record batch
# from_fasta_by_record leaves its tables in HBM (Profile.counts downloads on first access) while no more than this many bytes of
# them are alive; past it a batch is downloaded at once, as before round 6 (KPAL_DEVICE_PROFILE_BYTES; 0: always download)
_DEVICE_PROFILE_BYTES = int(os.environ.get('KPAL_DEVICE_PROFILE_BYTES', str(64 << 30)))


class _DeviceBatch(object):
    """The tables of one batch of ``from_fasta_by_record`` in device memory: n x 4^k int64, never written again.  Every
    profile of the batch that has not been asked for its counts yet holds a reference; the memory goes back when the last
    one is dropped or materialised."""
    live_bytes = 0

    _POOL_BYTES = 1 << 30      # freed allocations a context keeps for the next batch of the same size (hipMalloc + hipFree of a
                               # 128 MiB table cost a from_sequences call of a million reads 5 of its 23 ms)

    def __init__(self, ctx, nbytes, n_tables):
        self.ctx, self.nbytes, self.n_tables = ctx, int(nbytes), int(n_tables)
        pool = getattr(ctx, '_table_pool', None)
        if pool and pool.get(self.nbytes):
            self.ptr = pool[self.nbytes].pop()
            ctx._table_pool_bytes -= self.nbytes
        else:
            self.ptr = ctx.alloc(self.nbytes)
        self.host = None
        _DeviceBatch.live_bytes += self.nbytes

    def table(self, index):
        """Table ``index`` on the host.  The first request brings the WHOLE batch over in one copy (a profile at a time cost
        140 us per 512 KiB table: `kpal count --by-record` asks for every one of them); the rows are views of that array, as
        the tables of a batch were before round 6."""
        if self.host is None:
            host = np.empty((self.n_tables, self.nbytes // (8 * self.n_tables)), dtype=np.int64)
            self.ctx.d2h(host, self.ptr)
            self.host = host
        return self.host[index]

    def __del__(self):
        try:
            _DeviceBatch.live_bytes -= self.nbytes
            ctx = self.ctx
            if getattr(ctx, '_h', None):           # (a context that has been closed took its allocations with it)
                if getattr(ctx, '_table_pool', None) is None:
                    ctx._table_pool, ctx._table_pool_bytes = {}, 0
                if ctx._table_pool_bytes + self.nbytes <= _DeviceBatch._POOL_BYTES:
                    ctx._table_pool.setdefault(self.nbytes, []).append(self.ptr)
                    ctx._table_pool_bytes += self.nbytes
                else:
                    ctx.free(self.ptr)
        except Exception:       # pragma: no cover  (interpreter shutdown)
            pass

_FASTA_CHUNK = 64 << 20  # FASTA text read per device feed (cut back to a record boundary)
_DROPPED = {ord(' '): None, ord('\r'): None}   # removed from the joined record (Biopython's SimpleFastaParser)


def _fasta_records(handle):
    """Yield ``(name, sequence)`` per FASTA record, tokenised like ``Bio.SeqIO.parse(handle, 'fasta')``
    (what kpal/klib.py:111 delegates to): text before the first ``>`` is skipped, the name is the
    first word of the title, every sequence line is ``rstrip()``-ed and the joined record loses its
    spaces and ``\\r`` -- whitespace such as a tab INSIDE a line stays and separates k-mer windows."""
    name = None
    parts = []
    for line in handle:
        if isinstance(line, bytes):
            line = line.decode('ascii', 'replace')
        if line.startswith('>'):
            if name is not None:
                yield name, ''.join(parts).translate(_DROPPED)
            title = line[1:].strip()
            name = title.split(None, 1)[0] if title else ''
            parts = []
        elif name is not None:
            parts.append(line.rstrip())
    if name is not None:
        yield name, ''.join(parts).translate(_DROPPED)


def _plain_file(handle):
    """``(path, byte position)`` when ``handle`` is an ordinary file opened for reading whose raw bytes ARE its text and whose
    position is known -- an ``open(path)`` / ``open(path, 'rb')`` / ``argparse.FileType('r')`` handle on a regular file, in an
    ASCII-compatible encoding, not yet read from (text mode) or at any position (binary mode) -- else None (StringIO, pipes,
    gzip / bz2 wrappers, sockets, handles somebody has read from ...).  The library then reads the file itself
    (``kpal_count_feed_fasta_file``); ``/proc/self/fd/N`` names the open file whatever its path was."""
    raw = handle
    text_mode = isinstance(handle, io.TextIOWrapper)
    if text_mode:
        try:
            if codecs.lookup(handle.encoding or '').name not in ('utf-8', 'ascii', 'iso8859-1'):
                return None
        except LookupError:
            return None
        raw = handle.buffer
    if isinstance(raw, io.BufferedReader):
        raw = raw.raw
    if type(raw) is not io.FileIO or not raw.readable():
        return None
    try:
        fd = raw.fileno()
        if not stat.S_ISREG(os.fstat(fd).st_mode):
            return None
        position = handle.tell()
    except (OSError, ValueError):
        return None
    if text_mode and position != 0:      # (a text handle's tell() is an opaque cookie anywhere but at the start)
        return None
    path = '/proc/self/fd/%d' % fd
    return (path, position) if os.path.exists(path) else None


def _first_boundary(text):
    """Index of the end-of-line byte before the first header line of ``text`` (the first
    occurrence of EOL followed by ``>``), or -1; a chunk that itself starts with ``>`` starts a
    record at 0 only if the previous chunk ended with an EOL, which the caller's carry contains."""
    a = text.find(b'\n>')
    b = text.find(b'\r>', 0, a + 1) if a >= 0 else text.find(b'\r>')   # only an earlier one matters
    if a < 0:
        return b
    return a if b < 0 else min(a, b)


def _last_boundary(text, begin):
    """Index of the end-of-line byte before the last header line at or after ``begin``, or -1."""
    a = text.rfind(b'\n>', begin)
    b = text.rfind(b'\r>', max(begin, a))    # only a later one matters
    return max(a, b)


def _line_end(text, start):
    """Index of the first end-of-line byte (``\\n`` or ``\\r``) of ``text`` at or after ``start``, or ``len(text)``."""
    a = text.find(b'\n', start)
    if a < 0:
        a = len(text)
    b = text.find(b'\r', start, a)
    return a if b < 0 else b


def _whole_record_chunks(handle):
    """FASTA text of ``handle`` in pieces of about ``_FASTA_CHUNK`` bytes that hold WHOLE records: ``(bytes, str or None, encoding)`` --
    the text as bytes (what the device tokenises) and, for handles that yield ``str`` in an encoding of their own, the same
    text as the ``str`` the names are read from (one byte per character: positions agree)."""
    reader, keep_str, encoding = handle, False, 'ascii'
    if isinstance(handle, io.TextIOWrapper):
        try:                              # an ASCII-compatible text encoding: its bytes, undecoded (titles are decoded with it)
            if codecs.lookup(handle.encoding or '').name in ('utf-8', 'ascii', 'iso8859-1') and handle.tell() == 0:
                reader, encoding = handle.buffer, handle.encoding
        except (LookupError, OSError, ValueError):
            reader = handle
    pending = bytearray()                 # text not handed out yet: ends inside a record
    pending_str = []
    while True:
        text = reader.read(_FASTA_CHUNK)
        if not text:
            break
        if not isinstance(text, bytes):
            keep_str = True
            pending_str.append(text)
            text = text.encode('latin-1', 'replace')
        searched = max(len(pending) - 1, 0)   # (a boundary is two bytes: EOL + '>')
        pending += text
        cut = _last_boundary(pending, searched)
        if cut >= 0:                      # whole records up to the end of line before the last header line
            out = bytes(pending[:cut + 1])
            del pending[:cut + 1]
            out_str = None
            if keep_str:
                joined = ''.join(pending_str)
                out_str, pending_str = joined[:cut + 1], [joined[cut + 1:]]
            yield out, out_str, encoding
    if pending:
        yield bytes(pending), (''.join(pending_str) if keep_str else None), encoding


def _join_block(block):
    """One flat byte string for a block of sequences, ``\\n`` between them: C-speed joins for the
    homogeneous cases (all ``str`` / all ``bytes``), per-item encoding otherwise."""
    try:
        return '\n'.join(block).encode('latin-1', 'replace')
    except TypeError:
        pass
    try:
        return b'\n'.join(block)
    except TypeError:
        return b'\n'.join(_encode(s) for s in block)


def _gather_feed(ctx, sequences):
    """Feed ``sequences`` to the running count through the C gatherer: the items of a list are copied (each followed by the
    separator ``\\n``) into a page-locked buffer by several threads and handed to ``kpal_count_feed_pinned`` buffer by
    buffer.  A sequence is never split over two feeds (windows do not span feeds); one that does not fit the buffer goes
    through ``kpal_count_feed`` on its own; items the gatherer does not know (``str`` with characters beyond latin-1,
    ``memoryview`` ...) are encoded here, one at a time."""
    # one page-locked buffer per context, kept ON the context: it goes when the context is closed (kpal_ctx_destroy frees it)
    if getattr(ctx, '_gather_buffer', None) is None:
        ctx._gather_buffer = ctx.host_alloc(_GATHER_BYTES)
    address, view = ctx._gather_buffer
    cap = view.size
    fill = 0
    it = None if isinstance(sequences, (list, tuple)) else iter(sequences)
    while True:
        block = sequences if it is None else list(itertools.islice(it, 1 << 16))
        pos = 0
        while pos < len(block):
            pos, nbytes, status = _kpal_gather.gather(block, pos, address + fill, cap - fill, _GATHER_THREADS)
            fill += nbytes
            if status == 0:
                break
            data = None
            if status == 2:                      # an item the gatherer does not read: encoded here
                data = _encode(block[pos]) + b'\n'
                if len(data) <= cap - fill:
                    view[fill:fill + len(data)] = np.frombuffer(data, dtype=np.uint8)
                    fill += len(data)
                    pos += 1
                    continue
            if fill:                             # the buffer is full: hand it over, go on with an empty one
                ctx.count_feed_pinned(address, fill)
                fill = 0
            elif data is not None or len(block[pos]) + 1 > cap:
                ctx.count_feed(data if data is not None else _encode(block[pos]))   # larger than the buffer: on its own
                pos += 1
        if it is None or not block:
            break
    if fill:
        ctx.count_feed_pinned(address, fill)


def _encode(sequence):
    if isinstance(sequence, (bytes, bytearray, memoryview)):
        return bytes(sequence)
    # non-latin-1 characters can never be nucleotides: they become '?' (a separator)
    return str(sequence).encode('latin-1', 'replace')


class Profile(object):
    """A k-mer profile: ``counts[i]`` is the count of the k-mer whose 2-bit big-endian encoding
    (A=0, C=1, G=2, T=3) is ``i`` (kpal/klib.py:26-61).

    :arg counts: ``numpy.ndarray`` of length ``4**k`` (kept by reference, like the original).
    :arg str name: optional profile name (no ``/`` or ``.``).
    """

    _nucleotide_to_binary = {'A': 0, 'a': 0, 'C': 1, 'c': 1, 'G': 2, 'g': 2, 'T': 3, 't': 3}
    _binary_to_nucleotide = {0: 'A', 1: 'C', 2: 'G', 3: 'T'}

    def __init__(self, counts, name=None):
        self.length = int(math.log(len(counts), 4))
        self.counts = counts
        self.name = name

    # ---- counts that may still be in HBM -----------------------------------------------------
    # The reference's ``counts`` is a plain attribute holding a NumPy array that callers read AND write in place
    # (kpal/klib.py:58-61, kmer.py:137-146).  Here it is a property: a profile made by from_fasta_by_record starts with its table
    # in device memory only (``_device`` = (batch, index)); the first access downloads it, and from then on the host array is
    # the one truth -- the device copy is let go at that moment, because whoever holds the array may change it.  Distances,
    # matrices and the summaries of profiles nobody has looked at read the device copies (kdistlib.py, _device_stats).
    @classmethod
    def _from_device(cls, batch, index, length, name, private=False):
        """private: a copy() of a profile that is still in HBM -- it shares the (never written) device table, but when it is asked
        for counts it downloads its own array: the row of the batch's host array may be in someone's hands already."""
        p = cls.__new__(cls)
        p._counts, p._device = None, (batch, int(index), bool(private))
        p.length = int(length)
        p.name = name
        return p

    @property
    def counts(self):
        if self._counts is None and self._device is not None:
            batch, index, private = self._device
            if private:
                out = np.empty(4 ** self.length, dtype=np.int64)
                batch.ctx.d2h(out, batch.ptr + index * out.nbytes)
            else:
                out = batch.table(index)
            self._counts, self._device = out, None
        return self._counts

    @counts.setter
    def counts(self, value):
        self._counts, self._device = value, None

    # (pickling / copy.deepcopy of a profile whose table is still in HBM: the counts travel, the device handle does not)
    def __getstate__(self):
        return {'counts': self.counts, 'name': self._name, 'length': self.length}

    def __setstate__(self, state):
        self._counts, self._device = state['counts'], None
        self.length, self._name = state['length'], state['name']

    def _device_counts(self):
        """(context, device address) of the int64 table while it lives in HBM only, else None."""
        if self._counts is None and self._device is not None:
            batch, index, _ = self._device
            return batch.ctx, batch.ptr + index * 8 * 4 ** self.length
        return None

    # ---- constructors --------------------------------------------------------------------
    @classmethod
    def from_file(cls, handle, name=None):
        """Load from an open HDF5 k-mer file (kpal/klib.py:63-76)."""
        name = name or sorted(handle['profiles'].keys())[0]
        return cls(handle['profiles/' + name][:], name=name)

    @classmethod
    def from_file_old_format(cls, handle, name=None):
        """Load the pre-1.0 plaintext format: three header lines, then one count per line
        (kpal/klib.py:78-95)."""
        for _ in range(3):
            next(handle)
        return cls(np.loadtxt(handle, dtype='int64'), name=name)

    @classmethod
    def from_fasta(cls, handle, length, name=None):
        """One profile over all records of a FASTA handle (kpal/klib.py:97-112).

        The text is handed to the GPU in chunks of whole records; header removal, line joining
        and record separation happen on the device (``kpal_count_feed_fasta``), so there is no
        per-character or per-line Python work."""
        length = int(length)
        if length < 1 or length > _native.KPAL_MAX_K:
            raise ValueError('k-mer length must be in 1..%d (got %d)' % (_native.KPAL_MAX_K, length))
        ctx = _native.context()
        ctx.count_begin(length)
        plain = _plain_file(handle)
        if plain is not None:
            # an ordinary file: the library reads it itself -- parallel preads into pinned memory, H2D copy, flattening and
            # counting pipelined chunk by chunk; no byte of the text passes through Python
            ctx.count_feed_fasta_file(plain[0], plain[1], 0)
            handle.seek(0, os.SEEK_END)      # the handle has been consumed, as by the reference's SeqIO.parse loop
            return cls._finish_count(ctx, length, name)
        reader = handle
        if isinstance(handle, io.TextIOWrapper):
            try:                              # a pipe or a decompressing wrapper in an ASCII-compatible text encoding: its bytes, undecoded
                if codecs.lookup(handle.encoding or '').name in ('utf-8', 'ascii', 'iso8859-1') and handle.tell() == 0:
                    reader = handle.buffer
            except (LookupError, OSError, ValueError):
                reader = handle
        carry = b''   # the unfinished last record of the previous chunk (small)
        while True:
            text = reader.read(_FASTA_CHUNK)
            if not text:
                break
            if not isinstance(text, bytes):
                text = text.encode('latin-1', 'replace')
            view = np.frombuffer(text, dtype=np.uint8)   # zero-copy: slices below are views
            begin = 0
            if carry:
                # finish the straddling record: up to the first record boundary of this chunk
                first = _first_boundary(text)
                if first < 0:
                    carry += text
                    continue
                ctx.count_feed_fasta(carry + text[:first + 1])
                carry = b''
                begin = first + 1
            # whole records of this chunk; the (possibly unfinished) last one is carried over
            cut = _last_boundary(text, begin)
            if cut >= begin:
                ctx.count_feed_fasta(view[begin:cut + 1])
                carry = text[cut + 1:]
            else:
                carry = text[begin:]
        if carry:
            ctx.count_feed_fasta(carry)
        return cls._finish_count(ctx, length, name)

    @classmethod
    def from_fasta_by_record(cls, handle, length, prefix=None):
        """One profile per FASTA record, named by record (kpal/klib.py:114-133).

        The text is handed to the GPU in chunks of whole records: the records are found ON THE DEVICE
        (``kpal_fasta_records_begin``: the flattening kernels of ``from_fasta`` + an index of record starts and header
        lines) and counted in batches (``kpal_fasta_records_count``: one kernel launch and one table download for up to
        ``_RECORD_BATCH_BYTES`` of tables).  The host reads the HEADER lines only, for the names; there is no per-line or
        per-character Python work.  A k whose table alone exceeds that budget falls back to ``from_sequences`` per
        record."""
        length = int(length)
        if length < 1 or length > _native.KPAL_MAX_K:
            raise ValueError('k-mer length must be in 1..%d (got %d)' % (_native.KPAL_MAX_K, length))
        prefix = prefix + '_' if prefix else ''
        table_bytes = 8 * 4 ** length
        per_batch = _RECORD_BATCH_BYTES // table_bytes
        if per_batch < 2:
            for i, (record_name, seq) in enumerate(_fasta_records(handle)):
                yield cls.from_sequences([seq], length, name=prefix + (record_name or str(i + 1)))
            return
        ctx = _native.context()
        index = 0                                  # records seen so far (kpal/klib.py:132: the 1-based index names an untitled record)
        plain = _plain_file(handle)
        if plain is not None and os.path.getsize(plain[0]) > plain[1]:
            # an ordinary file: the library reads it itself (parallel preads into pinned memory, pieces of whole records);
            # the names are read from the header lines through a read-only mapping of the file
            import mmap
            encoding = handle.encoding if isinstance(handle, io.TextIOWrapper) else 'ascii'
            # The reference's generators are independent of each other (zip() of two, nesting, one abandoned half-way), and
            # this one yields with a piece indexed on the process-wide context.  So nothing of the scan lives in the context
            # across a yield: the scan is opened at `resume`, asked for ONE piece and closed again; and when another scan has
            # used the context between two batches of a piece, the piece -- bytes [at, resume) -- is indexed again.
            def index_piece(begin, end):
                ctx.fasta_records_file_open(plain[0], begin, end)
                try:
                    piece = ctx.fasta_records_file_next()
                    return piece, (ctx.fasta_records_file_tell() if piece is not None else end)
                finally:
                    ctx.fasta_records_file_close()

            with open(plain[0], 'rb') as raw, mmap.mmap(raw.fileno(), 0, access=mmap.ACCESS_READ) as text:
                resume = plain[1]
                while True:
                    piece, resume = index_piece(resume, 0)
                    if piece is None:
                        break
                    n_records, _, at = piece
                    if n_records == 0:
                        continue
                    scan = ctx._records_scan
                    header_off, _ = ctx.fasta_records_index()
                    names = []
                    for h in (header_off + np.uint64(at)).tolist():
                        words = text[h + 1:_line_end(text, h)].decode(encoding, 'replace').split(None, 1)
                        index += 1
                        names.append(prefix + (words[0] if words else str(index)))
                    for first in range(0, n_records, per_batch):
                        if ctx._records_scan != scan:
                            again, _ = index_piece(at, resume)
                            if again is None or again[0] != n_records:
                                raise RuntimeError('from_fasta_by_record: bytes %d..%d of %s changed under the scan' % (at, resume, plain[0]))
                            scan = ctx._records_scan
                        n = min(per_batch, n_records - first)
                        for profile in cls._record_batch(ctx, length, first, n, names):
                            yield profile
            handle.seek(0, os.SEEK_END)          # the handle has been consumed, as by the reference's SeqIO.parse loop
            return
        for text, text_str, encoding in _whole_record_chunks(handle):
            n_records, _ = ctx.fasta_records_begin(text)
            if n_records == 0:
                continue
            header_off, _ = ctx.fasta_records_index()
            names = []
            for h in header_off.tolist():
                end = _line_end(text, h)
                title = text_str[h + 1:end] if text_str is not None else text[h + 1:end].decode(encoding, 'replace')
                words = title.split(None, 1)
                index += 1
                names.append(prefix + (words[0] if words else str(index)))
            scan = ctx._records_scan
            for first in range(0, n_records, per_batch):
                if ctx._records_scan != scan:        # another by-record scan used the context since the last batch: this text again
                    ctx.fasta_records_begin(text)
                    scan = ctx._records_scan
                n = min(per_batch, n_records - first)
                for profile in cls._record_batch(ctx, length, first, n, names):
                    yield profile

    @classmethod
    def _finish_count(cls, ctx, length, name):
        """The finished count of ``ctx`` as a profile.  The table stays in HBM -- copied device-to-device out of the context's
        count table, which the next count reuses, into an allocation of its own -- until something asks for ``counts``
        (`kpal count` saving it: the one download it always was; a distance between two freshly counted profiles: none at all);
        past the budget of live device tables it is downloaded at once."""
        nbytes = 8 * 4 ** length
        if _DEVICE_PROFILE_BYTES and _DeviceBatch.live_bytes + nbytes <= _DEVICE_PROFILE_BYTES:
            ctx.count_finish(to_host=False)
            table, _ = ctx.count_table()
            batch = _DeviceBatch(ctx, nbytes, 1)
            ctx.d2d(batch.ptr, table, nbytes)
            ctx.sync()
            return cls._from_device(batch, 0, length, name)
        return cls(ctx.count_finish(), name=name)

    @classmethod
    def _record_batch(cls, ctx, length, first, n, names):
        """The profiles of records [first, first + n) of the text the context has indexed.  Their tables stay in HBM (one
        allocation per batch) while the budget of live device tables allows; else they are downloaded at once."""
        nbytes = n * 8 * 4 ** length
        if _DEVICE_PROFILE_BYTES and _DeviceBatch.live_bytes + nbytes <= _DEVICE_PROFILE_BYTES:
            batch = _DeviceBatch(ctx, nbytes, n)
            ctx.fasta_records_count_device(length, first, n, batch.ptr)
            return [cls._from_device(batch, j, length, names[first + j]) for j in range(n)]
        tables = ctx.fasta_records_count(length, first, n)
        return [cls(tables[j], name=names[first + j]) for j in range(n)]

    @classmethod
    def from_sequences(cls, sequences, length, name=None):
        """Count all k-mers of every sequence (kpal/klib.py:135-170) on the GPU.

        Windows never span two sequences or a character outside ``AaCcGgTt``.
        """
        length = int(length)
        if length < 1 or length > _native.KPAL_MAX_K:
            raise ValueError('k-mer length must be in 1..%d (got %d)' % (_native.KPAL_MAX_K, length))
        ctx = _native.context()
        ctx.count_begin(length)
        if _kpal_gather is not None:
            _gather_feed(ctx, sequences)
            return cls._finish_count(ctx, length, name)
        it = iter(sequences)
        pending = []
        size = 0
        while True:
            block = list(itertools.islice(it, _JOIN_BLOCK))
            if not block:
                break
            data = _join_block(block)
            pending.append(data)
            size += len(data) + 1
            if size >= _FEED_BYTES:
                # feeds are independent: a window never spans two feeds, nor two sequences
                ctx.count_feed(pending[0] if len(pending) == 1 else b'\n'.join(pending))
                pending, size = [], 0
        if pending:
            ctx.count_feed(pending[0] if len(pending) == 1 else b'\n'.join(pending))
        return cls._finish_count(ctx, length, name)

    # ---- properties ------------------------------------------------------------------------
    @property
    def name(self):
        return self._name

    @name.setter
    def name(self, name):
        if name and ('/' in name or '.' in name):
            raise ValueError('Profile name may not contain / or . characters.')
        self._name = name

    @property
    def number(self):
        """Number of possible k-mers of this length."""
        return 4 ** self.length if self._counts is None and self._device is not None else len(self.counts)

    @property
    def non_zero(self):
        dev = self._device_counts()
        return int(dev[0].stats_device(dev[1], self.number).non_zero) if dev else np.count_nonzero(self.counts)

    @property
    def total(self):
        dev = self._device_counts()
        return np.int64(dev[0].stats_device(dev[1], self.number).total) if dev else self.counts.sum()

    def _device_stats(self):
        """``kpal_stats`` over integer counts (one upload -- none for a table that is still in HBM --, two streaming passes and
        a radix select), or None when the counts are not integers (scaled profiles keep NumPy's float formulation)."""
        dev = self._device_counts()
        if dev:
            return dev[0].stats_device(dev[1], self.number)
        c = np.asanyarray(self.counts)
        if c.dtype.kind not in 'iub' or c.ndim != 1 or c.size == 0:
            return None
        return _native.context().stats(c)

    @property
    def mean(self):
        """Mean count (kpal/klib.py:206-211)."""
        s = self._device_stats()
        return self.counts.mean() if s is None else np.float64(s.mean)

    @property
    def median(self):
        """Median count (kpal/klib.py:213-218), exact: radix select on the device."""
        s = self._device_stats()
        return np.median(self.counts) if s is None else np.float64(s.median)

    @property
    def std(self):
        """Population standard deviation of the counts (kpal/klib.py:220-225)."""
        s = self._device_stats()
        return self.counts.std() if s is None else np.float64(s.std)

    def summary(self):
        """``total``, ``non_zero``, ``mean``, ``median`` and ``std`` at once: one ``kpal_stats`` call for integer
        counts (the attributes ``save`` writes and ``kpal info`` prints), the properties otherwise."""
        s = self._device_stats()
        if s is None:
            return dict((key, getattr(self, key)) for key in ('total', 'non_zero', 'mean', 'median', 'std'))
        return {'total': np.int64(s.total), 'non_zero': int(s.non_zero), 'mean': np.float64(s.mean),
                'median': np.float64(s.median), 'std': np.float64(s.std)}

    # ---- I/O -------------------------------------------------------------------------------
    def save(self, handle, name=None):
        """Write to an open HDF5 k-mer file, dataset ``profiles/<name>`` with the summary
        attributes of format 1.0.0 (kpal/klib.py:227-256, doc/fileformat.rst:23-46)."""
        if name and ('/' in name or '.' in name):
            raise ValueError('Profile name may not contain / or . characters.')
        name = name or self.name or next(str(n) for n in itertools.count(1) if str(n) not in handle['profiles'])
        dataset = handle.create_dataset('profiles/' + name, data=self.counts, dtype='int64', compression='gzip')
        attrs = self.summary()        # all five summaries from one pass over the counts
        dataset.attrs['length'] = self.length
        for key in ('total', 'non_zero', 'mean', 'median', 'std'):
            dataset.attrs[key] = attrs[key]
        handle.flush()
        return name

    def copy(self):
        """Deep copy (kpal/klib.py:258-267).  (A table that is still in HBM is never written: the copy shares it until either
        profile is asked for its counts.)"""
        if self._counts is None and self._device is not None:
            return type(self)._from_device(self._device[0], self._device[1], self.length, self.name, private=True)
        return type(self)(self.counts.copy(), name=self.name)

    def merge(self, profile, merger=metrics.mergers['sum']):
        """Merge another profile into this one with a vectorised merger (kpal/klib.py:269-283)."""
        code = metrics.merger_code(merger)
        l, r = np.asanyarray(self.counts), np.asanyarray(profile.counts)
        if code is not None and l.dtype.kind in 'iub' and r.dtype.kind in 'iub' and l.shape == r.shape and l.ndim == 1:
            self.counts = _native.context().merge(l, r, code)     # built-in merger: one HIP kernel
        else:
            self.counts = merger(self.counts, profile.counts)     # user callable / float profiles: as the reference

    # ---- hot path: balance / split ------------------------------------------------------------
    def _rc_index(self):
        """rc(i) for every i (vectorised form of reverse_complement, kpal/klib.py:394-412)."""
        k = self.length
        i = np.arange(4 ** k, dtype=np.uint64)
        rc = np.zeros_like(i)
        comp = ~i
        for d in range(k):
            rc |= ((comp >> np.uint64(2 * d)) & np.uint64(3)) << np.uint64(2 * (k - 1 - d))
        return rc.astype(np.int64)

    def balance(self):
        """``counts[i] += counts[rc(i)]`` for every k-mer, palindromes doubled, in place
        (kpal/klib.py:285-298) -- one HIP kernel for integer counts.  Non-integer counts (a scaled
        profile) cannot enter the integer kernel: they take the same sums as one NumPy expression, in
        the dtype of ``counts`` like the reference's loop."""
        c = np.asanyarray(self.counts)
        if c.dtype.kind not in 'iub':
            c[...] = c + c[self._rc_index()]
            return
        if c.dtype == np.int64 and c.flags['C_CONTIGUOUS'] and c.flags['WRITEABLE']:
            _native.context().balance_inplace(c, self.length)
        else:
            tmp = np.ascontiguousarray(c, dtype=np.int64).copy()
            _native.context().balance_inplace(tmp, self.length)
            self.counts[...] = tmp

    def split(self):
        """Doubled forward / reverse-complement halves in ascending k-mer order
        (kpal/klib.py:300-327) -- ``kpal_split``; NumPy for non-integer counts."""
        c = np.asanyarray(self.counts)
        if c.dtype.kind not in 'iub':
            rc = self._rc_index()
            i = np.arange(c.size)
            lower, pal = i < rc, i == rc
            keep = lower | pal
            forward = np.where(pal, c, c * 2)[keep]
            reverse = np.where(pal, c, c[rc] * 2)[keep]
            return forward, reverse
        return _native.context().split(c, self.length)

    # ---- container helpers ----------------------------------------------------------------------
    def shrink(self, factor=1):
        """Reduce k by ``factor`` by summing groups of ``4**factor`` bins (kpal/klib.py:329-352)."""
        if self.length <= factor:
            raise ValueError('Reduction factor should be smaller than k-mer size.')
        if factor < 0:    # the reference fails in range() with a float step (4 ** factor)
            raise TypeError("'float' object cannot be interpreted as an integer")
        c = np.asanyarray(self.counts)
        if factor == 0:   # groups of one: a fresh int64 copy
            self.counts = np.array(c, dtype='int64')
            return
        if c.dtype.kind in 'iub':
            self.counts = _native.context().shrink(c, self.length, factor)
        else:   # the reference builds int64 counts from whatever it holds (np.fromiter(..., dtype='int64'))
            self.counts = c.reshape(-1, 4 ** factor).sum(axis=1).astype('int64')
        self.length -= factor

    def shuffle(self):
        """Randomise the profile in place (kpal/klib.py:354-358)."""
        np.random.shuffle(self.counts)

    def dna_to_binary(self, sequence):
        """DNA string -> integer; ``KeyError`` on a non-ACGT character (kpal/klib.py:360-375)."""
        result = 0
        for nucleotide in sequence:
            result = (result << 2) | self._nucleotide_to_binary[nucleotide]
        return result

    def binary_to_dna(self, number):
        """Integer -> DNA string of this profile's k (kpal/klib.py:377-392)."""
        letters = []
        for _ in range(self.length):
            letters.append(self._binary_to_nucleotide[number & 3])
            number >>= 2
        return ''.join(reversed(letters))

    def reverse_complement(self, number):
        """Reverse complement in the binary representation (kpal/klib.py:394-412)."""
        return _native.reverse_complement(number, self.length)

    def _ratios_matrix(self):
        """All relative frequencies count[i]/count[j]/total, -1.0 where count[j] is 0
        (kpal/klib.py:414-437)."""
        c = np.asarray(self.counts, dtype='float64')
        total = float(self.total)
        with np.errstate(divide='ignore', invalid='ignore'):
            m = (c[:, None] / c[None, :]) / total
        m[:, c == 0] = -1.0
        return m.tolist()

    def _freq_diff_matrix(self):
        """All |count[i]-count[j]|/total, 0 where count[j] is 0 (kpal/klib.py:439-456)."""
        c = np.asarray(self.counts)
        total = self.total
        m = np.abs(c[:, None] - c[None, :]) / total
        out = m.tolist()
        for j in np.nonzero(c == 0)[0]:
            for row in out:
                row[j] = 0
        return out

    def print_counts(self):
        """Print ``<k-mer> <count>`` lines (kpal/klib.py:458-463)."""
        for i in range(self.number):
            print(self.binary_to_dna(i), self.counts[i])

    def _print_ratios(self, ratios):
        """Print a ratios matrix (kpal/klib.py:465-487)."""
        print((self.length + 1) * ' ', end=' ')
        for i in range(self.number):
            print(self.binary_to_dna(i), end='   ')
        print()
        for i in range(self.number):
            print(self.binary_to_dna(i), end=' ')
            for j in range(self.number):
                print('{{0:.{0}f}}'.format(self.length).format(ratios[i][j]), end=' ')
            print()
