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
"""
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
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
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
