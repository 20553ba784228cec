#!/usr/bin/env python
"""The in-library RCCL path on ONE GPU (tests/test_gpu_dist.py::test_library_rccl_world_1; run on the GPU box):
communicator of world size 1 from kpal_comm_unique_id, kpal_comm_reduce_table serial and pipelined + balance over three
steps each, merged tables against oracle.balance(oracle.count); kpal_comm_max_f64; errors after kpal_comm_destroy."""
import os
import sys

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

import oracle
from kpal_amd import _native

ctx = _native.Context(_native.default_device())
ctx.comm_init(0, 1, _native.comm_unique_id())
for k in (12, 13):
    for pipelined in (False, True):
        wants, gots = [], []
        for step in range(3):
            buf = oracle.synth_reads(80 + step, 0, 20000, 150, noisy=True)
            ctx.count_begin(k, 'partition2_quads' if k == 13 else 'auto')
            ctx.count_feed(buf)
            ctx.comm_reduce_table(0, balance=True, pipelined=pipelined)
            wants.append(oracle.balance(oracle.count_flat(buf, k), k))
            if pipelined and step == 1:
                # two reduces in flight at most: the merged table of step 0 is still intact while step 1's is being reduced
                ctx.count_begin(k)
                ctx.count_feed(buf[:1000])
            ctx.sync()
            ptr, bins = ctx.comm_merged_table()
            got = np.empty(bins, dtype=np.int64)
            ctx.d2h(got, ptr)
            gots.append(got)
        for g, w in zip(gots, wants):
            assert np.array_equal(g, w), (k, pipelined)
# the bin-range merge (kpal_comm_reduce_scatter_table: ncclReduceScatter + the mirrored-range exchange; world 1: the one range is
# the whole table, the rank's own block a device copy) + kpal_comm_gather_table, and the two permutation kernels alone at every
# power-of-two "world" -- one process playing all ranks: every rank's range of the balanced table against the oracle
for k in (9, 12, 13):
    buf = oracle.synth_reads(90 + k, 0, 30000, 150, noisy=True)
    want_plain = oracle.count_flat(buf, k)
    want = oracle.balance(want_plain, k)
    for balance in (True, False):
        ctx.count_begin(k, 'partition2_quads' if k == 13 else 'auto')
        ctx.count_feed(buf)
        ctx.comm_reduce_scatter_table(balance=balance)
        ctx.sync()
        ptr, first, bins = ctx.comm_merged_range()
        assert (first, bins) == (0, 4 ** k)
        got = np.empty(bins, dtype=np.int64)
        ctx.d2h(got, ptr)
        assert np.array_equal(got, want if balance else want_plain), (k, balance)
        ctx.comm_gather_table()
        ctx.sync()
        assert ctx.comm_merged_range()[1:] == (0, 4 ** k)
    # pack / unpack kernels at world 2, 4, 8: rank r's blocks to rank q are exchanged on the host
    tab = ctx.alloc(8 * 4 ** k)
    for world in (2, 4, 8):
        n1 = 4 ** k // world
        n2 = n1 // world
        sends = []
        d_send = ctx.alloc(8 * n1)
        for r in range(world):
            ctx.h2d(tab, want_plain)
            ctx.range_pack_device(k, r, world, tab, d_send)
            h = np.empty(n1, dtype=np.int64)
            ctx.d2h(h, d_send)
            sends.append(h)
        for r in range(world):
            recv = np.concatenate([sends[q][r * n2:(r + 1) * n2] for q in range(world)])
            ctx.h2d(d_send, recv)
            ctx.h2d(tab, want_plain)
            ctx.range_unpack_device(k, r, world, tab, d_send)
            out = np.empty(4 ** k, dtype=np.int64)
            ctx.d2h(out, tab)
            assert np.array_equal(out[r * n1:(r + 1) * n1], want[r * n1:(r + 1) * n1]), (k, world, r)
        ctx.free(d_send)
    ctx.free(tab)
# distance matrix from bin-range shards (world size 1: the one shard is the whole range; the split / all-reduce / join of the
# per-pair partials runs all the same): LDS-staged kernels (64-bin multiples) and the register-tile kernel (a ragged range)
rs = np.random.RandomState(3)
for k, P, bins in ((8, 12, 4 ** 8), (7, 9, 4 ** 7 - 100)):
    prof = rs.poisson(3.0, (P, 4 ** k)).astype(np.int64)
    prof[1, ::5] = 0
    sl = np.ascontiguousarray(prof[:, :bins])
    d = ctx.alloc(sl.nbytes)
    ctx.h2d(d, sl)
    for metric, name in ((0, 'prod'), (1, 'sum'), (2, 'euclidean')):
        got = ctx.comm_distance_matrix_device(P, bins, d, metric)
        if bins == 4 ** k:
            want = oracle.distance_matrix_values(prof, k, False, name)
        else:       # a ragged range is not a profile: pair by pair through the oracle's metric functions
            want = np.array([oracle.euclidean(sl[i], sl[j]) if metric == 2 else oracle.multiset(sl[i], sl[j], name)
                             for i in range(1, P) for j in range(i)])
        if metric == 2:
            assert np.array_equal(got, want), (k, name)
        else:
            assert np.max(np.abs(got - want) / np.abs(want)) <= 1e-9, (k, name)
    ctx.free(d)
assert ctx.comm_max(3.5) == 3.5
ctx.comm_destroy()
try:
    ctx.comm_reduce_table(0)
except RuntimeError:
    pass
else:
    raise AssertionError('kpal_comm_reduce_table without a communicator must fail')
ctx.close()
print('RCCL_LIBRARY_OK')
