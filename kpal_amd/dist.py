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
