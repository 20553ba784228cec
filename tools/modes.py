#!/usr/bin/env python
"""Developer probe: is the run-to-run spread of the k=12 kernels a property of the process (memory
placement) or of the moment (clocks)?  Re-creates the context several times inside one process."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kpal_amd import _native

reads = 40_000_000
for cycle in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    ctx = _native.Context(0)
    d = ctx.alloc(reads * 151)
    ctx.synth_reads_device(2, 0, reads, 150, d)
    out = []
    for it in range(4):
        if it == 1:
            ctx.prof_enable(True); ctx.prof_reset()
        ctx.count_begin(12)
        ctx.count_feed_device(d, reads * 151)
        ctx.count_finish(to_host=False)
    prof = ctx.prof_get()
    print('cycle %d: chunk_scatter %.3f ms each, chunk_hist %.3f ms each, input at 0x%x' % (
        cycle, prof['chunk_scatter'][0] / prof['chunk_scatter'][1], prof['chunk_hist'][0] / prof['chunk_hist'][1], d), flush=True)
    ctx.free(d)
    ctx.close()
    del ctx
