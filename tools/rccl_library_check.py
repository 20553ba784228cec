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
