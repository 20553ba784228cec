#!/usr/bin/env python
"""ONE RANK of a world of W processes that share GPU 0 and run the library's kpal_comm_* protocol against each other
(tests/test_gpu_dist.py::test_library_comm_world_over_fake_rccl starts W of these with KPAL_RCCL_LIBRARY pointing at the
stand-in built from tests/native/fake_rccl.cpp -- RCCL itself refuses two ranks on one device).  Everything but the transport is
the real thing: every rank has its own context, streams, tables and kernels.

    python tests/rccl_world_rank.py RANK WORLD ID_FILE

What is checked, every result against the oracle on the reads of ALL ranks:
  * kpal_comm_reduce_table to rank 0 + balance, serial and pipelined, three steps each, k = 12 (one-level) and 13 (two-level);
  * kpal_comm_reduce_scatter_table (ncclReduceScatter + the mirrored-range exchange, W * (W - 1) messages in one group) with
    and without balance at k = 9, 12, 13: every rank checks ITS range; then kpal_comm_gather_table: every rank checks the table;
  * kpal_comm_distance_matrix_device from bin-range shards (the ranks hold different ranges -- one of them too short for the LDS-staged
    kernels -- and must agree on the kernel family), all three metrics; one rank with invalid arguments: an error on EVERY rank, nobody hangs;
  * kpal_comm_max_f64.
"""
import os
import sys
import time

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

import oracle
from kpal_amd import _native

rank, world, id_file = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ctx = _native.Context(0)
if rank == 0:
    ident = _native.comm_unique_id()
    with open(id_file + '.tmp', 'wb') as fh:
        fh.write(ident)
    os.rename(id_file + '.tmp', id_file)
else:
    _native.comm_probe()
    t0 = time.time()
    while not os.path.exists(id_file):
        if time.time() - t0 > 120:
            sys.exit('rank %d: no id file' % rank)
        time.sleep(0.05)
    with open(id_file, 'rb') as fh:
        ident = fh.read()
ctx.comm_init(rank, world, ident)
N = 12000                                               # reads per rank and step


def shard(seed, r):
    return oracle.synth_reads(seed, r * N, N, 150, noisy=True)


def download(ptr, bins):
    out = np.empty(bins, dtype=np.int64)
    ctx.d2h(out, ptr)
    return out


# ---- whole-table reduce to rank 0
for k in (12, 13):
    strategy = 'partition2_quads' if k == 13 else 'auto'
    for pipelined in (False, True):
        wants, gots = [], []
        for step in range(3):
            seed = 100 + 10 * k + step
            ctx.count_begin(k, strategy)
            ctx.count_feed(shard(seed, rank))
            ctx.comm_reduce_table(0, balance=True, pipelined=pipelined)
            if pipelined and step == 1:
                # the next count starts while the reduce is in flight (two at most): the merged table of step 1 must not see it
                ctx.count_begin(k)
                ctx.count_feed(shard(seed, rank)[:1000])
            ctx.sync()
            if rank == 0:
                gots.append(download(*ctx.comm_merged_table()))
                wants.append(oracle.balance(oracle.count_flat(b''.join(bytes(shard(seed, r)) for r in range(world)), k), k))
        for g, w in zip(gots, wants):
            assert np.array_equal(g, w), ('reduce', k, pipelined)
# ---- bin-range merge
for k in (9, 12, 13):
    seed = 200 + k
    plain = oracle.count_flat(b''.join(bytes(shard(seed, r)) for r in range(world)), k)
    balanced = oracle.balance(plain, k)
    for balance in (True, False):
        want = balanced if balance else plain
        ctx.count_begin(k, 'partition2_quads' if k == 13 else 'auto')
        ctx.count_feed(shard(seed, rank))
        ctx.comm_reduce_scatter_table(balance=balance)
        ctx.sync()
        ptr, first, bins = ctx.comm_merged_range()
        assert (first, bins) == (rank * (4 ** k // world), 4 ** k // world), (first, bins)
        assert np.array_equal(download(ptr, bins), want[first:first + bins]), ('range', k, balance, rank)
        ctx.comm_gather_table()
        ctx.sync()
        ptr, first, bins = ctx.comm_merged_range()
        assert (first, bins) == (0, 4 ** k)
        assert np.array_equal(download(ptr, bins), want), ('gather', k, balance, rank)
# ---- distance matrix from bin-range shards: equal 64-bin multiples (LDS-staged kernels on every rank), then ranges of which the
# last is too short for them (one rank cannot take the staged kernels: all must take the plain ones)
rs = np.random.RandomState(3)
k, P = 8, 10
prof = rs.poisson(3.0, (P, 4 ** k)).astype(np.int64)
prof[1, ::5] = 0
n = 4 ** k
step = (n - 2048) // max(world - 1, 1) // 64 * 64
small_last = [r * step for r in range(world - 1)] + [n - 2048, n] if world > 1 else [0, n]   # the last rank's 2048 bins are too few for the staged kernels
for cuts in (np.linspace(0, n, world + 1).astype(int), small_last):
    lo, hi = int(cuts[rank]), int(cuts[rank + 1])
    sl = np.ascontiguousarray(prof[:, lo:hi])
    d = ctx.alloc(sl.nbytes)
    ctx.h2d(d, sl)
    for metric, name in ((0, 'prod'), (1, 'sum'), (2, 'euclidean')):
        got = ctx.comm_distance_matrix_device(P, hi - lo, d, metric)
        want = oracle.distance_matrix_values(prof, k, False, name)
        if metric == 2:
            assert np.array_equal(got, want), ('matrix', name, rank)
        else:
            assert np.max(np.abs(got - want) / np.abs(want)) <= 1e-9, ('matrix', name, rank)
    # the last rank passes an empty slice: every rank gets the error, none waits in a collective
    try:
        ctx.comm_distance_matrix_device(P, 0 if rank == world - 1 else hi - lo, d, 0)
    except (RuntimeError, ValueError):
        pass
    else:
        raise AssertionError('an invalid slice on one rank must fail on every rank')
    ctx.free(d)
assert ctx.comm_max(rank + 0.5) == world - 0.5
ctx.comm_destroy()
ctx.close()
print('RCCL_WORLD_OK rank %d of %d' % (rank, world), flush=True)
