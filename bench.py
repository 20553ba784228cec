#!/usr/bin/env python
"""bench.py -- Gbases/s of k-mer counting (k=12, 150 bp synthetic reads) on N MI355X.

A "step" is one pass of the hot path over this rank's resident batch of synthetic reads:
zero the 4^k table, count every k-mer (kpal_count_feed_device), [N>1: one RCCL reduce of the
int64 tables to rank 0], balance the table (Profile.balance).  Inputs are generated on the
device before the timed region (HBM-resident); weak scaling: every rank holds --reads reads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--k 12]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (see DESIGN.md section 6 for the field definitions).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured achievable)
PMC_PROFILE = os.path.join(ROOT, 'profiles', 'r1', 'pmc_hbm_traffic.json')


def pmc_traffic(kernel, k, input_bytes_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 +
    WRITE_SIZE, gfx950 correction applied), scaled to this run's bytes per launch.  PMC counters
    cannot be read from inside the bench process; None if no profile of this kernel is committed."""
    try:
        path = PMC_PROFILE if k == 12 else PMC_PROFILE.replace('.json', '_k%d.json' % k)
        with open(path) as fh:
            prof = json.load(fh)
        for name, rec in prof['kernels'].items():
            if ('::%s_kernel' % kernel) in name and ('<%d' % k) in name:
                return rec['hbm_bytes_per_dispatch_corrected'] * input_bytes_per_launch / prof['input_bytes_per_launch_avg']
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(k, read_len, budget_reads):
    """Time the CPU oracle (C port of kpal/klib.py:149-170) on a bounded sample of the same
    workload, 1 thread and all cores.  Reported baseline, never the target."""
    import oracle
    cores = min(os.cpu_count() or 1, 64)     # private 4^k histograms per thread: cap the fan-out
    buf = oracle.synth_reads(2, 0, budget_reads, read_len)
    t0 = time.perf_counter()
    c1 = oracle.count_flat(buf, k, threads=1)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    cn = oracle.count_flat(buf, k, threads=cores)
    tn = time.perf_counter() - t0
    assert int(c1.sum()) == int(cn.sum()) == budget_reads * (read_len - k + 1)
    bases = budget_reads * read_len
    return {
        'value': bases / t1 / 1e9, 'unit': 'Gbases/s', 'cores': 1, 'kind': 'port',
        'sample': '%d synthetic %d bp reads, k=%d, oracle/kpal_oracle.c (1 thread: %.2f s)' % (budget_reads, read_len, k, t1),
        'all_cores': {'value': bases / tn / 1e9, 'cores': cores, 'seconds': tn},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--reads', type=int, default=100_000_000, help='reads per GPU (weak scaling)')
    ap.add_argument('--read-len', type=int, default=150)
    ap.add_argument('--k', type=int, default=12)
    ap.add_argument('--strategy', default='auto')
    ap.add_argument('--cpu-reads', type=int, default=4_000_000, help='reads in the CPU-baseline sample')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling: --reads is the TOTAL, split over the GPUs (default: weak, --reads per GPU)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run (one process per GPU)' % args.gpus)
        args.gpus = world

    import torch
    import torch.distributed as td
    from kpal_amd import _native, dist as kdist

    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU (no CPU fallback for the hot path)')
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        td.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    ctx = _native.Context(local_rank)
    k, L = args.k, args.read_len
    seed = 2 if world == 1 else 3
    if args.strong:
        first_read, n_reads = kdist.shard_range(args.reads, rank, world)   # fixed total, contiguous shards
        total_reads = args.reads
    else:
        n_reads = args.reads
        first_read = rank * n_reads                     # shard s = reads [s*R, (s+1)*R)
        total_reads = world * n_reads
    nbytes = n_reads * (L + 1)
    dev_buf = ctx.alloc(nbytes)
    ctx.synth_reads_device(seed, first_read, n_reads, L, dev_buf)
    ctx.sync()

    ctx.count_begin(k, args.strategy)                   # allocates the table once
    table = kdist.table_as_tensor(ctx) if world > 1 else None
    table_ptr, bins = ctx.count_table()

    def step():
        ctx.count_begin(k, args.strategy)               # zero the 4^k table
        ctx.count_feed_device(dev_buf, nbytes)
        if world > 1:
            ctx.sync()                                  # table complete before RCCL touches it
            kdist.reduce_counts(table, dst=0)
            torch.cuda.current_stream().synchronize()
        if rank == 0:
            ctx.balance_device(k, table_ptr)            # Profile.balance on the merged table
        ctx.sync()

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.prof_enable(True)
    ctx.prof_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = ctx.prof_get()
    ctx.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the last step's merged + balanced table has exactly 2 * (#k-mers) counts
    ok = True
    if rank == 0:
        import numpy as np
        out = np.empty(bins, dtype=np.int64)
        ctx.d2h(out, table_ptr)
        ok = int(out.sum()) == 2 * total_reads * (L - k + 1)

    if rank == 0:
        steps = max(args.steps, 1)
        bases_per_step = total_reads * L
        ms_per_step = elapsed / steps * 1e3
        value = bases_per_step / (elapsed / steps) / 1e9
        # roofline of the dominant kernel (HIP events on the launch stream, this rank)
        alg_bytes_step = n_reads * (L + 1) + 8 * bins    # B_in + 8*4^k, per GPU
        kern = {n: v for n, v in prof.items() if v[1] > 0}
        dom = max(kern, key=lambda n: kern[n][0]) if kern else None
        roofline = None
        if dom:
            tot_ms, launches = kern[dom]
            per_launch_bytes = alg_bytes_step * steps / launches
            avg_ms = tot_ms / launches
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': achieved / HBM_PEAK_GBS,
                        'traffic': pmc_traffic(dom, k, n_reads * (L + 1) * steps / launches),
                        'traffic_source': 'profiles/r1/pmc_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes), scaled per launch',
                        'avg_launch_ms': avg_ms,
                        'launches_per_step': launches / steps,
                        'algorithmic_bytes_per_launch': per_launch_bytes,
                        'pipeline_frac': alg_bytes_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        'kernels_ms_per_step': {n: v[0] / steps for n, v in sorted(kern.items())}}
        line = {
            'metric': 'Gbases/s k-mer counted (k=%d, %dbp synthetic)' % (k, L), 'value': value, 'unit': 'Gbases/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None, 'dtype': 'int64', 'data': 'synthetic',
            'config': {'workload': 'k=%d, %d synthetic %dbp reads per GPU resident in HBM, count%s+balance'
                                   % (k, n_reads, L, '+RCCL reduce' if world > 1 else ''),
                       'k': k, 'reads_per_gpu': n_reads, 'read_len': L, 'strategy': args.strategy,
                       'parallelism': 'reads sharded x%d, 1 reduce(int64 sum)' % world},
            'checksum_ok': ok,
            'roofline': roofline,
        }
        if world == 1 and not args.no_cpu:
            line['cpu_baseline'] = cpu_baseline(k, L, args.cpu_reads)
        print(json.dumps(line), flush=True)

    ctx.free(dev_buf)
    ctx.close()
    if world > 1:
        td.destroy_process_group()
    if rank == 0 and not ok:
        sys.exit('checksum mismatch')


if __name__ == '__main__':
    main()
