"""Multi-GPU counting: one process per GPU, shards of reads, one RCCL reduce of the count
vectors over xGMI.

Counting shards embarrassingly: every read -- more generally every piece of the
input cut at a separator -- is counted independently and the 4^k int64 count vectors add
("merging ... is equivalent to first concatenating both fasta files", doc/tutorial.rst:94-95;
``metrics.mergers['sum']``, kpal/metrics.py:175).  So each rank counts its contiguous block of
reads into a private device table and the only collective is ONE
``torch.distributed.reduce(SUM, int64)`` of 4^k elements to rank 0 (backend ``nccl`` == RCCL;
integer addition makes the result bit-exact for any ring/tree order).  The reference itself has
no parallelism of any kind.

FASTA input (BASELINE north_star: "input FASTA shards partitioned embarrassingly across the 8 GPUs of one node
with a single RCCL reduce"): ``fasta_shards`` cuts one or several FASTA files into ``world`` byte ranges of about equal size --
at record boundaries where there is one near the ideal cut, else INSIDE a record at a line start (or, in a line longer than
64 KiB, anywhere), in which case the right-hand range carries the k - 1 bases before the cut as a read-only halo: every k-mer
window is counted by exactly one rank.  ``count_fasta_sharded`` counts a rank's ranges (``kpal_count_feed_fasta_file``: the
library reads the file itself), the tables are merged by the one reduce.  ``python -m kpal_amd.dist count ...`` (under
``torch.distributed.run``, or alone) is the runnable entry: one profile over all input files, written by rank 0.
"""
import collections
import mmap
import os

import numpy as np


def shard_range(n_units, rank, world_size):
    """Contiguous, balanced block of ``n_units`` for ``rank``: (first, count).
    Blocks differ in size by at most one and tile [0, n_units) exactly."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError('bad rank/world_size %r/%r' % (rank, world_size))
    base, extra = divmod(int(n_units), int(world_size))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def reduce_counts(table, dst=0, group=None):
    """Sum per-rank count vectors onto ``dst`` with one collective.

    ``table``: torch int64 tensor (a CUDA tensor under ``nccl``/RCCL; a CPU tensor under ``gloo``
    in the CPU tests).  Reduced in place; only ``dst`` holds the full sum afterwards."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.reduce(table, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return table


def table_as_tensor(ctx):
    """Zero-copy torch view of the context's device count table (for the RCCL reduce)."""
    import torch
    view = ctx.count_table_view()
    return torch.as_tensor(view, device='cuda:%d' % ctx.device)


def count_synth_sharded(ctx, k, seed, n_reads_total, read_len, rank, world_size, dev_buf=None, strategy='auto'):
    """Count this rank's block of the synthetic read set into the context's device
    table and return (first_read, n_reads).  The caller then reduces ``table_as_tensor(ctx)``."""
    first, n = shard_range(n_reads_total, rank, world_size)
    nbytes = n * (read_len + 1)
    own = dev_buf is None
    if own:
        dev_buf = ctx.alloc(max(nbytes, 16))
    try:
        ctx.synth_reads_device(seed, first, n, read_len, dev_buf)
        ctx.count_begin(k, strategy)
        ctx.count_feed_device(dev_buf, nbytes)
        ctx.count_finish(to_host=False)
    finally:
        if own:
            ctx.free(dev_buf)
    return first, n


# ----------------------------------------------------------------------------------------------------------------------
# The bin-range merge (k >= 13): reduce-scatter + mirrored-range exchange.  The library does it with RCCL on its own stream
# (kpal_comm_reduce_scatter_table); this is the torch.distributed variant of the same protocol, with the same index arithmetic
# (csrc/range_index.hpp, restated here in NumPy for CPU tensors: the gloo tests).
# ----------------------------------------------------------------------------------------------------------------------
def _revcomp(idx, k):
    """Profile.reverse_complement (kpal/klib.py:394-412) on a uint64 array."""
    x = ~idx.astype(np.uint64)
    for sh, m in ((2, 0x3333333333333333), (4, 0x0F0F0F0F0F0F0F0F), (8, 0x00FF00FF00FF00FF), (16, 0x0000FFFF0000FFFF)):
        m = np.uint64(m)
        x = ((x >> np.uint64(sh)) & m) | ((x & m) << np.uint64(sh))
    x = (x >> np.uint64(32)) | (x << np.uint64(32))
    return x >> np.uint64(64 - 2 * k)


def range_merge_supported(k, world):
    """RangeIndex::valid() of csrc/range_index.hpp: a power-of-two world of 2^w ranks with 2 ceil(w / 2) <= k -- whole BASES
    of the reverse complement name the destination rank, so an odd w needs one bit more than 4^k >= world^2 says (k = 3 with 8
    ranks and k = 5 with 32 are refused)."""
    w = int(world).bit_length() - 1
    return world >= 1 and (1 << w) == world and 1 <= k <= 31 and 2 * ((w + 1) // 2) <= k


def range_geometry(k, world):
    """-> (w, bins per rank, bins per pair of ranks) of the bin-range merge; see range_merge_supported."""
    w = int(world).bit_length() - 1
    if world < 1 or (1 << w) != world:
        raise ValueError('the bin-range merge needs a power-of-two number of ranks (got %d)' % world)
    if not range_merge_supported(k, world):
        raise ValueError('the bin-range merge needs 2 ceil(log2(ranks) / 2) <= k (k=%d, %d ranks)' % (k, world))
    n1 = 4 ** k >> w
    return w, n1, n1 >> w


def range_owner_pos(j, k, world):
    """RangeIndex::owner / pos of csrc/range_index.hpp for a uint64 array of table indices -> (owner rank, position in its block)."""
    w, n1, _ = range_geometry(k, world)
    j = j.astype(np.uint64)
    owner = (j >> np.uint64(2 * k - w)) if w else np.zeros_like(j)
    lb = 2 * ((w + 1) // 2)
    spare = lb - w
    mid = (j & np.uint64(n1 - 1)) >> np.uint64(lb)
    if spare:
        sub = (_revcomp(j, k) >> np.uint64(2 * k - lb)) & np.uint64(1)
        return owner, (mid << np.uint64(1)) | sub
    return owner, mid


def range_pack(table, k, rank, world):
    """NumPy restatement of range_pack_kernel: this rank's range of ``table`` (int64[4^k]) -> int64[4^k / world], block q = what rank q gets."""
    w, n1, n2 = range_geometry(k, world)
    j = np.arange(rank * n1, (rank + 1) * n1, dtype=np.uint64)
    q, _ = range_owner_pos(_revcomp(j, k), k, world)
    _, p = range_owner_pos(j, k, world)
    send = np.empty(n1, dtype=np.int64)
    send[(q * np.uint64(n2) + p).astype(np.int64)] = table[rank * n1:(rank + 1) * n1]
    return send


def range_unpack(table, recv, k, rank, world):
    """NumPy restatement of range_unpack_kernel: balances this rank's range of ``table`` in place with the received blocks."""
    w, n1, n2 = range_geometry(k, world)
    i = np.arange(rank * n1, (rank + 1) * n1, dtype=np.uint64)
    j = _revcomp(i, k)
    q, p = range_owner_pos(j, k, world)
    table[rank * n1:(rank + 1) * n1] += recv[(q * np.uint64(n2) + p).astype(np.int64)]


def reduce_scatter_balance(table, k, balance=True, ctx=None, group=None):
    """The bin-range merge over torch.distributed: ``table`` (torch int64[4^k]: this rank's counts; a CUDA tensor viewing the
    context's table under nccl / RCCL -- then ``ctx`` runs the library's pack / unpack kernels -- or a CPU tensor under gloo) ->
    this rank's range of the merged (and balanced) table, in place: ONE reduce_scatter of 4^k int64 + ONE all_to_all of 4^k / W.
    Returns ``(first_bin, n_bins)``."""
    import torch
    import torch.distributed as td
    on = td.is_available() and td.is_initialized()
    rank = td.get_rank(group) if on else 0
    world = td.get_world_size(group) if on else 1
    w, n1, n2 = range_geometry(k, world)
    mine = table[rank * n1:(rank + 1) * n1]
    if world > 1:
        td.reduce_scatter_tensor(mine, table, op=td.ReduceOp.SUM, group=group)      # in place: the output is the input's own block
    if balance:
        if table.is_cuda:
            if ctx is None:
                raise ValueError('a CUDA table needs the context whose kernels pack and unpack it')
            send = torch.empty(n1, dtype=torch.int64, device=table.device)
            recv = torch.empty(n1, dtype=torch.int64, device=table.device)
            torch.cuda.synchronize()                           # (torch's stream -> the context's)
            ctx.range_pack_device(k, rank, world, table.data_ptr(), send.data_ptr())
            ctx.sync()
            if world > 1:
                td.all_to_all_single(recv, send, group=group)
            else:
                recv.copy_(send)
            torch.cuda.synchronize()
            ctx.range_unpack_device(k, rank, world, table.data_ptr(), recv.data_ptr())
            ctx.sync()
        else:
            send = torch.from_numpy(range_pack(table.numpy(), k, rank, world))
            recv = torch.empty_like(send)
            if world > 1:
                td.all_to_all_single(recv, send, group=group)
            else:
                recv.copy_(send)
            range_unpack(table.numpy(), recv.numpy(), k, rank, world)
    return rank * n1, n1


class TableReducer(object):
    """The multi-GPU merge step of the counting path: per-rank count tables -> their sum on rank 0
    (+ ``balance`` there), one collective per step.

    ``table``   torch int64 tensor viewing this rank's count table (``table_as_tensor``; a CPU tensor in
                the gloo tests).
    ``sync``    callable that waits for the counting kernels (the context's stream is not torch's).
    ``balance`` callable(tensor) run on the merged table on rank 0, or None.
    ``mode``    'int64' (default): ``reduce(SUM)`` of the table in place, 128 MiB per rank at k = 12.
                'u32': the counts travel as 32-bit words -- half the bytes over xGMI (a ring reduce is
                bound by the per-link rate) -- whenever the LARGEST per-rank bin times the number of ranks
                fits 31 bits (one scalar all-reduce(MAX) decides, identically on every rank); otherwise
                that step falls back to int64.  Integer sums: bit-exact either way.
    ``overlap`` False (default): reduce, then balance, inside the step.  True: the table is copied to one of
                TWO side buffers and reduced asynchronously while the NEXT step counts into the table; the
                reduce of step i is waited for (and balanced) at the start of step i+1's reduce, the last
                one by ``drain()``.  The buffers alternate: the balanced result of step i (``result()``,
                balanced asynchronously on the context's stream) stays untouched while step i+1's table is
                copied into the other one, and is only overwritten by step i+2's copy -- after the
                ``sync()`` that opens ``reduce_step`` i+2 has waited for that balance.  ``result()`` is valid
                from ``drain()`` / the next ``reduce_step`` until the ``reduce_step`` after that.
                Costs two extra tables of HBM.

    (bench.py's default multi-GPU path is the in-library RCCL reduce, ``Context.comm_*``; this class is the
    torch.distributed variant and the one the gloo CPU tests drive.)"""

    def __init__(self, table, sync=None, balance=None, mode='int64', overlap=False, dst=0, group=None):
        if mode not in ('int64', 'u32'):
            raise ValueError('mode must be int64 or u32')
        self.table = table
        self.sync = sync or (lambda: None)
        self.balance = balance
        self.mode = mode
        self.overlap = bool(overlap)
        self.dst = dst
        self.group = group
        self._bufs = [None, None]  # overlap / u32: side buffers the merged counts end up in (alternating)
        self._turn = 0
        self._pending = None      # (work, tensor32 or None)
        self._result = table
        self.steps_u32 = 0
        self.steps_int64 = 0

    # -- helpers ---------------------------------------------------------------------------------
    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def _torch_sync(self):
        if self.table.is_cuda:
            import torch
            torch.cuda.current_stream().synchronize()

    def _fits_32(self, world):
        """Same answer on every rank: max over ranks of the largest (and smallest) bin."""
        import torch
        import torch.distributed as dist
        m = torch.stack([self.table.max(), -self.table.min()])
        if world > 1:
            dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        hi, neg_lo = int(m[0]), int(m[1])
        return neg_lo <= 0 and hi * world < 2 ** 31

    def _side_buffer(self):
        """The side buffer whose previous content (the result of two steps ago) may be overwritten."""
        import torch
        self._turn ^= 1
        if self._bufs[self._turn] is None:
            self._bufs[self._turn] = torch.empty_like(self.table)
        return self._bufs[self._turn]

    def _start(self):
        """Launch the collective for the current table; returns (work or None, source tensor, is32)."""
        import torch
        import torch.distributed as dist
        world, _ = self._world()
        use32 = self.mode == 'u32' and self._fits_32(world)
        if use32:
            src = self.table.to(torch.int32)
            self.steps_u32 += 1
        elif self.overlap:
            src = self._side_buffer()
            src.copy_(self.table)
            self.steps_int64 += 1
        else:
            src = self.table
            self.steps_int64 += 1
        work = None
        if world > 1:
            work = dist.reduce(src, dst=self.dst, op=dist.ReduceOp.SUM, group=self.group, async_op=self.overlap)
        return work, src, use32

    def _finish(self, work, src, use32):
        import torch
        _, rank = self._world()
        if work is not None:
            work.wait()
        if use32:
            buf = self._side_buffer()
            if rank == self.dst:
                buf.copy_(src)                # int32 -> int64
            self._result = buf
        else:
            self._result = src
        self._torch_sync()                     # the context's stream may touch the result now
        if rank == self.dst and self.balance is not None:
            self.balance(self._result)

    # -- the step --------------------------------------------------------------------------------
    def reduce_step(self):
        """Call after the step's counting kernels have been launched."""
        self.sync()                            # the table is complete before torch / RCCL read it
        if not self.overlap:
            self._finish(*self._start())
            return
        self.drain()                           # the previous step's reduce: wait, widen, balance
        self._pending = self._start()
        self._torch_sync()                     # the copy out of the table is done: the next count may zero it

    def drain(self):
        if self._pending is not None:
            pending, self._pending = self._pending, None
            self._finish(*pending)

    def result(self):
        """The merged (and balanced) table of the last finished step; valid on ``dst``."""
        return self._result

    def result_ptr(self):
        return self._result.data_ptr()


# ----------------------------------------------------------------------------------------------------------------------
# FASTA shards
# ----------------------------------------------------------------------------------------------------------------------
FastaSegment = collections.namedtuple('FastaSegment', 'path begin end prefix')
FastaSegment.__doc__ = """Bytes ``[begin, end)`` of the FASTA file ``path``; ``prefix``: text that logically precedes them (``b''``, or
``b'>\\n'`` + the k - 1 sequence bytes before a cut inside a record)."""

_LONG_LINE = 64 << 10     # a cut inside a longer line does not move back to the line's start
_BLANKS = frozenset(b' \t\x0b\x0c\x1c\x1d\x1e\x1f\x85\xa0')   # what the flattening drops at the end of a line (fasta_kernels.hpp)


def _line_start(mm, o):
    """Start of the line that contains byte ``o`` (lines end at ``\\n`` or ``\\r``)."""
    a = mm.rfind(b'\n', 0, o)
    b = mm.rfind(b'\r', max(a, 0), o)      # only a later one matters
    return max(a, b) + 1


def _last_header(mm, before):
    """Start of the last header line (``>`` at a line start) that begins before ``before``, or -1."""
    a = mm.rfind(b'\n>', 0, before)
    b = mm.rfind(b'\r>', max(a, 0), before)
    at = max(a, b)
    if at >= 0:
        return at + 1
    return 0 if before > 0 and mm[0:1] == b'>' else -1


def _flatten_fragment(raw):
    """Sequence bytes of a fragment of ONE record's sequence lines (no header inside), as the device flattening and
    ``klib._fasta_records`` produce them: every line right-stripped, spaces and line ends removed."""
    text = raw.decode('latin-1')
    return ''.join(line.rstrip() for line in text.replace('\r', '\n').split('\n')).replace(' ', '').encode('latin-1')


def _cut(mm, o, k):
    """A cut of the text near byte ``o`` (0 < o < len(mm)): -> (offset, prefix) with offset <= o.  The offset is the start of a
    header line (prefix empty: a record boundary), or lies inside a record -- at a line start, or for lines longer than 64 KiB
    wherever ``o`` fell (moved left past blanks) -- and then ``prefix`` is ``b'>\\n'`` + the last k - 1 sequence bytes of that
    record before the offset: the right-hand range counts the windows that reach across the cut, the left-hand one only those
    that end before it."""
    ls = _line_start(mm, o)
    if mm[ls:ls + 1] == b'>':
        return ls, b''
    at = ls
    if o - ls > _LONG_LINE:
        at = o
        while at > ls and mm[at - 1] in _BLANKS:
            at -= 1
    hdr = _last_header(mm, at)
    if hdr < 0:
        return at, b''                       # text before the first header: ignored by whoever reads it
    # the record's sequence starts after its header line
    e = hdr
    n = len(mm)
    while e < n and mm[e] not in (10, 13):
        e += 1
    seq_start = min(e + 1, n)
    if at <= seq_start or k <= 1:
        return at, b'>\n'
    want = k - 1
    window = 4096
    while True:
        lo = max(seq_start, at - window)     # (a fragment may begin inside a line: only line ENDS are stripped)
        flat = _flatten_fragment(mm[lo:at])
        if len(flat) >= want or lo <= seq_start:
            return at, b'>\n' + flat[-want:]
        window *= 4


def fasta_shards(paths, world, k):
    """Cut the FASTA files ``paths`` (their concatenation, in order) into ``world`` shards of about equal size:
    -> ``world`` lists of :class:`FastaSegment`.  Counting every segment of every shard and adding the tables gives the
    profile of all records of all files (kpal/klib.py:97-112 on each file, merged with ``sum``: doc/tutorial.rst:94-95)."""
    if isinstance(paths, (str, bytes, os.PathLike)):
        paths = [paths]
    paths = [os.fspath(p) for p in paths]
    if world < 1:
        raise ValueError('bad world size %r' % (world,))
    k = int(k)
    sizes = [os.path.getsize(p) for p in paths]
    total = sum(sizes)
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    # the cut points: (file index, offset, prefix), the first at the very beginning, the last at the very end
    cuts = [(0, 0, b'')]
    for r in range(1, world):
        c = r * total // world
        f = int(np.searchsorted(starts, c, side='right')) - 1
        f = min(max(f, 0), len(paths) - 1)
        o = c - int(starts[f])
        if o <= 0 or sizes[f] == 0:
            cuts.append((f, 0, b''))
            continue
        with open(paths[f], 'rb') as fh:
            mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            try:
                off, prefix = _cut(mm, min(o, sizes[f] - 1), k)
            finally:
                mm.close()
        cuts.append((f, off, prefix))
    cuts.append((len(paths) - 1, sizes[-1] if sizes else 0, b''))
    for i in range(1, len(cuts)):            # moving a cut back to a line start must not pass the cut before it
        if (cuts[i][0], cuts[i][1]) < (cuts[i - 1][0], cuts[i - 1][1]):
            cuts[i] = cuts[i - 1]
    shards = []
    for r in range(world):
        (f0, o0, prefix), (f1, o1, _) = cuts[r], cuts[r + 1]
        segs = []
        for f in range(f0, f1 + 1):
            begin = o0 if f == f0 else 0
            end = o1 if f == f1 else sizes[f]
            if end > begin:
                segs.append(FastaSegment(paths[f], begin, end, prefix if f == f0 else b''))
        shards.append(segs)
    return shards


def count_fasta_sharded(ctx, k, segments, strategy='auto'):
    """Count one rank's shard (``fasta_shards(...)[rank]``) into the context's device table: ``kpal_count_begin`` + one
    ``kpal_count_feed_fasta_file`` per segment.  The caller then merges the tables (``Context.comm_reduce_table``, or
    ``reduce_counts(table_as_tensor(ctx))``)."""
    ctx.count_begin(k, strategy)
    for seg in segments:
        ctx.count_feed_fasta_file(seg.path, seg.begin, seg.end, seg.prefix)


def segment_text(seg):
    """The FASTA text a segment stands for (its prefix + its bytes of the file): what a rank counts (tests, small inputs)."""
    with open(seg.path, 'rb') as fh:
        fh.seek(seg.begin)
        return seg.prefix + fh.read(seg.end - seg.begin)


def profile_from_fasta_sharded(paths, k, name=None, root=0):
    """One :class:`kpal_amd.klib.Profile` over all records of the FASTA files ``paths``, counted by all ranks of the current
    ``torch.distributed`` job (one process per GPU; without an initialised process group: this process alone): shards by
    ``fasta_shards``, one reduce(SUM) of the int64 tables to ``root``.  Returns the profile on ``root``, None elsewhere."""
    from . import _native, klib
    try:
        import torch.distributed as td
        grouped = td.is_available() and td.is_initialized()
    except ImportError:
        grouped = False
    rank, world = (td.get_rank(), td.get_world_size()) if grouped else (0, 1)
    ctx = _native.context()
    count_fasta_sharded(ctx, k, fasta_shards(paths, world, k)[rank])
    if grouped:
        ctx.count_finish(to_host=False)
        table = table_as_tensor(ctx)
        ctx.sync()
        reduce_counts(table, dst=root)
        import torch
        torch.cuda.current_stream().synchronize()
    if rank != root:
        return None
    return klib.Profile(ctx.count_finish(), name=name)


def main(argv=None):
    """``python -m kpal_amd.dist count [-k K] [--name NAME] IN.fa [IN.fa ...] OUT.k`` -- `kpal count` over all GPUs of the node:
    start it with ``python -m torch.distributed.run --nproc-per-node N -m kpal_amd.dist count ...`` (one process per GPU), or
    plainly for one GPU.  ONE profile over all input files is written (kpal count writes one per file; merging them with
    `kpal merge` gives this profile, doc/tutorial.rst:94-95)."""
    import argparse
    ap = argparse.ArgumentParser(prog='python -m kpal_amd.dist', description=main.__doc__)
    sub = ap.add_subparsers(dest='command', required=True)
    c = sub.add_parser('count', help='count the k-mers of FASTA files, sharded over the ranks')
    c.add_argument('-k', dest='size', type=int, default=9, help='k-mer size (default: %(default)s)')      # kmer.py:789
    c.add_argument('--name', default=None, help='profile name (default: the first file\'s base name)')
    c.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                   help='torch.distributed backend of the merge (default: %(default)s = RCCL; gloo moves the tables through host memory)')
    c.add_argument('--device', type=int, default=None,
                   help='GPU of this rank (default: LOCAL_RANK; several ranks may share one GPU with --backend gloo -- RCCL refuses that)')
    c.add_argument('inputs', nargs='+', metavar='INPUT', help='FASTA files')
    c.add_argument('output', metavar='OUTPUT', help='k-mer profile file (HDF5)')
    args = ap.parse_args(argv)
    grouped = 'WORLD_SIZE' in os.environ and 'RANK' in os.environ      # started by torch.distributed.run (any world size)
    if grouped:
        import torch
        import torch.distributed as td
        local = args.device if args.device is not None else int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        if args.device is not None:
            os.environ['KPAL_DEVICE'] = str(local)   # the library's default context of this process: the same device
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            td.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            td.init_process_group('gloo')
    name = args.name or os.path.splitext(os.path.basename(args.inputs[0]))[0]
    profile = profile_from_fasta_sharded(args.inputs, args.size, name=name)
    if profile is not None:
        from . import files
        handle = files.ProfileFileType('w')(args.output)
        profile.save(handle)
        handle.close()
    if grouped:
        td.barrier()
        td.destroy_process_group()
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
