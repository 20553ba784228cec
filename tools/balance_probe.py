"""Time kpal_balance_device on a resident 4^k table: python3 tools/balance_probe.py K [reps]"""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kpal_amd import _native

k = int(sys.argv[1]) if len(sys.argv) > 1 else 15
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _native.Context(_native.default_device())
n = 2_000_000
d = ctx.alloc(n * 151)
ctx.synth_reads_device(5, 0, n, 150, d)
ctx.count_begin(k)
ctx.count_feed_device(d, n * 151)
ctx.count_finish(to_host=False)
ptr, bins = ctx.count_table()
ctx.balance_device(k, ptr)
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.balance_device(k, ptr)
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print('balance k=%d: %.3f ms  (%.2f TB/s for 16 * 4^k bytes)' % (k, dt * 1e3, 16.0 * bins / dt / 1e12))
