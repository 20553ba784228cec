import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch, torch.distributed as td
from kpal_amd import _native, dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
td.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
ctx = _native.Context(0)
first, n = dist.count_synth_sharded(ctx, 9, 3, 20000, 150, 0, 1)
t = dist.table_as_tensor(ctx)
print('tensor', t.dtype, t.shape, t.device, int(t.sum().item()), 'expected', 20000 * 142)
td.all_reduce(t)   # exercises RCCL on the zero-copy view
td.reduce(t, dst=0)
host = ctx.count_finish()
assert int(host.sum()) == 20000 * 142 == int(t.sum().item())
print('NCCL_VIEW_OK')
td.destroy_process_group()
