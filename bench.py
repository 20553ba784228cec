#!/usr/bin/env python
"""bench.py -- Gbases/s of k-mer counting (k=12, 150 bp synthetic reads) on N MI355X.

Default workload (BASELINE.json metric, config 2 / 3): a "step" is one pass of the hot path over this
rank's resident batch of synthetic reads: zero the 4^k table, count every k-mer
(kpal_count_feed_device), [N>1: one RCCL reduce of the count tables to rank 0], balance the table
(Profile.balance).  Inputs are generated on the device before the timed region (HBM-resident); weak
scaling: every rank holds --reads reads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--k 12]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--workload matrix (BASELINE config 5, one GPU): 64 profiles at k = 12 (profile p = 2 M reads, seed 100+p)
resident in HBM, a step = one kdistlib.distance_matrix value computation (kpal_distance_matrix_device).

Rank 0 prints ONE JSON line (see DESIGN.md section 5 for the field definitions).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured achievable)
FP64_VALU_PEAK_T = 39.3    # fp64 vector lane-instructions/s: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (78.6 TFLOP/s with FMA = half the guide's 157.3 TF fp32 vector rate)
FP64_MFMA_PEAK_T = 78.6    # fp64 matrix TFLOP/s (v_mfma_f64_16x16x4_f64: 2048 flop per 64 SIMD-cycles x 1024 SIMDs x 2.4 GHz)
PROFILE_DIR = os.path.join(ROOT, 'profiles', 'r2')


def source_sha():
    """sha256 over the kernel sources: ties a committed counter profile to the code that produced it
    (the GPU box has no .git, so a commit hash cannot be checked there)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'kpal_amd', 'csrc')
    for name in sorted(os.listdir(csrc)) + ['../../include/kpal_hip.h']:
        with open(os.path.join(csrc, name), 'rb') as fh:
            h.update(name.encode())
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, k, input_bytes_per_launch):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 +
    WRITE_SIZE, gfx950 correction applied), scaled to this run's input bytes per launch.  PMC counters
    cannot be read from inside the bench process, so this is a PROFILE of the same command, not a
    measurement of this run: it is only reported when the profile was taken from the same kernel
    sources (src_sha), else None with a note."""
    info = {'traffic': None, 'traffic_source': None, 'traffic_profile_head': None, 'traffic_profile_src_sha': None}
    path = os.path.join(PROFILE_DIR, 'pmc_hbm_traffic%s.json' % ('' if k == 12 else '_k%d' % k))
    try:
        with open(path) as fh:
            prof = json.load(fh)
    except (OSError, ValueError):
        info['traffic_source'] = 'no committed counter profile for k=%d' % k
        return info
    info['traffic_profile_head'] = prof.get('head')
    info['traffic_profile_src_sha'] = prof.get('src_sha')
    here = source_sha()
    if prof.get('src_sha') != here:
        info['traffic_source'] = '%s was taken from other kernel sources (src_sha %s, now %s): not reported' % (
            os.path.relpath(path, ROOT), prof.get('src_sha'), here)
        return info
    for name, rec in prof['kernels'].items():
        if ('::%s_kernel' % kernel) in name and ('<%d' % k) in name:
            info['traffic'] = rec['hbm_bytes_per_dispatch_corrected'] * input_bytes_per_launch / prof['input_bytes_per_launch_avg']
            info['traffic_source'] = '%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on the same kernel sources, scaled per launch)' % os.path.relpath(path, ROOT)
            return info
    info['traffic_source'] = '%s has no row for %s' % (os.path.relpath(path, ROOT), kernel)
    return info


def cpu_baseline(k, read_len, budget_reads):
    """CPU figures next to the GPU number (reported baselines, never the target):
    the oracle's C port (kpal/klib.py:149-170 restated) on a bounded sample of the same workload with 1
    thread and with all host cores (one shared table, relaxed atomic adds), and the pure-Python
    restatement of the reference's loop on BASELINE config 1 (the reference's own speed class)."""
    import oracle
    from oracle import pyref
    cores = os.cpu_count() or 1
    buf = oracle.synth_reads(2, 0, budget_reads, read_len)
    oracle.count_flat(buf[:151 * 1000], k)                      # page the table in outside the timed region
    t0 = time.perf_counter()
    c1 = oracle.count_flat(buf, k, threads=1)
    t1 = time.perf_counter() - t0
    # all cores: one shared table, relaxed atomic adds; on a many-socket host fewer threads can be faster (cache-line
    # ping-pong), so a few thread counts are timed and the best is reported with its own core count
    tn, best_threads, tried = None, cores, {}
    for threads in sorted(set([cores, min(cores, 64), min(cores, 16)]), reverse=True):
        t0 = time.perf_counter()
        cn = oracle.count_flat(buf, k, threads=threads)
        t = time.perf_counter() - t0
        tried[str(threads)] = t
        assert int(cn.sum()) == budget_reads * (read_len - k + 1)
        if tn is None or t < tn:
            tn, best_threads = t, threads
    assert int(c1.sum()) == budget_reads * (read_len - k + 1)
    bases = budget_reads * read_len
    # BASELINE config 1: 10 k reads, k = 9 through the interpreter loop
    reads1 = [bytes(r).decode() for r in oracle.synth_reads(1, 0, 10000, 150).reshape(-1, 151)[:, :150]]
    t0 = time.perf_counter()
    cp = pyref.from_sequences(reads1, 9)
    tp = time.perf_counter() - t0
    assert int(cp.sum()) == 1420000
    return {
        'value': bases / t1 / 1e9, 'unit': 'Gbases/s', 'cores': 1, 'kind': 'port',
        'sample': '%d synthetic %d bp reads, k=%d, oracle/kpal_oracle.c (1 thread: %.2f s)' % (budget_reads, read_len, k, t1),
        'all_cores': {'value': bases / tn / 1e9, 'cores': best_threads, 'host_cores': cores, 'seconds': tn,
                      'per_thread_efficiency': (bases / tn) / (bases / t1) / best_threads,
                      'seconds_by_threads': tried,
                      'note': 'one shared 4^k table, relaxed atomic adds (oracle/kpal_oracle.c); best of the thread counts tried'},
        'python_reference_loop': {'value': 1.5e6 / tp / 1e9, 'unit': 'Gbases/s', 'cores': 1, 'seconds': tp,
                                  'sample': 'BASELINE config 1 (10000 reads, k=9) through oracle/pyref.py, the statement-by-statement restatement of kpal/klib.py:149-170'},
    }


def matrix_workload(args):
    """BASELINE config 5 on one GPU: value = Gterms/s of the 64 x 64 lower triangle (2016 pairs x 4^12 bins)."""
    import numpy as np
    from kpal_amd import _native
    import oracle
    ctx = _native.Context(0)
    k, P, n = args.k, args.profiles, 4 ** args.k
    metric = {'prod': 0, 'sum': 1, 'euclidean': 2}[args.metric]
    nbytes = args.profile_reads * 151
    d = ctx.alloc(nbytes)
    dprof = ctx.alloc(P * n * 8)
    host = []
    for p in range(P):
        ctx.synth_reads_device(100 + p, 0, args.profile_reads, 150, d)
        ctx.count_begin(k)
        ctx.count_feed_device(d, nbytes)
        ctx.count_finish(to_host=False)
        ptr, _ = ctx.count_table()
        if p < 8:
            c = np.empty(n, dtype=np.int64)
            ctx.d2h(c, ptr)
            host.append(c)
            ctx.h2d(dprof + p * n * 8, c)
        else:
            c = np.empty(n, dtype=np.int64)
            ctx.d2h(c, ptr)
            ctx.h2d(dprof + p * n * 8, c)
    ctx.free(d)
    for _ in range(args.warmup):
        vals = ctx.distance_matrix_device(P, k, dprof, metric, args.balance)
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vals = ctx.distance_matrix_device(P, k, dprof, metric, args.balance)
    ctx.sync()
    elapsed = time.perf_counter() - t0
    prof = {n_: v for n_, v in ctx.prof_get().items() if v[1] > 0}
    ctx.prof_enable(False)
    steps = max(args.steps, 1)
    pairs = P * (P - 1) // 2
    terms = pairs * n
    ms = elapsed / steps * 1e3
    dom = max(prof, key=lambda n_: prof[n_][0])
    dom_ms = prof[dom][0] / prof[dom][1]
    mem_bytes = 8 * P * n * (2 if args.balance else 1)
    if metric == 2 and dom.startswith('gram'):
        # 10 of the 16 tile pairs of a 64-profile block are computed; 2 flop per multiply-add
        flops = 2.0 * n * sum(10 * 256 for _ in range((P + 63) // 64)) + 2.0 * n * 16 * 256 * ((P + 63) // 64) * ((P + 63) // 64 - 1) / 2
        roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': flops / (dom_ms * 1e-3) / 1e12, 'peak': FP64_MFMA_PEAK_T, 'unit': 'TFLOP/s',
                    'frac': flops / (dom_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_T, 'traffic': None,
                    'memory_frac': mem_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': mem_bytes,
                    'avg_launch_ms': dom_ms}
    else:
        # multiset term with staged reciprocals (matrix_accumulate_prod_rcp): sub, mul, fma = 3 fp64 slots, + per 16 terms of a
        # register tile 8 int -> double conversions and ~22 integer instructions for the term-count bytes: ~5 slots per term
        # (sum metric / int64 euclidean without the staged reciprocals: ~12)
        slots = 5.0 if metric == 0 else 12.0
        roofline = {'bound': 'fp64-valu', 'kernel': dom, 'achieved': terms * slots / (dom_ms * 1e-3) / 1e12, 'peak': FP64_VALU_PEAK_T,
                    'unit': 'Tinstr/s (fp64 lane-instructions; %.0f issue slots per term)' % slots,
                    'frac': terms * slots / (dom_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_T, 'traffic': None,
                    'memory_frac': mem_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': mem_bytes,
                    'avg_launch_ms': dom_ms}
    roofline['kernels_ms_per_step'] = {n_: v[0] / steps for n_, v in sorted(prof.items())}
    line = {
        'metric': 'Gterms/s distance matrix (%d profiles k=%d, %s%s)' % (P, k, args.metric, ' balanced' if args.balance else ''),
        'value': terms / (elapsed / steps) / 1e9, 'unit': 'Gterms/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64' if metric != 2 else 'int64/f64-mfma', 'data': 'synthetic',
        'config': {'workload': 'BASELINE config 5: %d profiles k=%d (%d synthetic 150bp reads each, seed 100+p) resident in HBM, kdistlib.distance_matrix %s%s' % (
            P, k, args.profile_reads, args.metric, ' +balance' if args.balance else ''), 'pairs': pairs},
        'roofline': roofline,
    }
    # parity spot-check + CPU baseline on an 8-profile subset
    if not args.no_cpu:
        t0 = time.perf_counter()
        want = oracle.distance_matrix_values(host, k, args.balance, args.metric)
        tc = time.perf_counter() - t0
        sub = np.array([vals[i * (i - 1) // 2 + j] for i in range(1, 8) for j in range(i)])
        rel = float(np.max(np.abs(sub - want) / np.maximum(np.abs(want), 1e-300)))
        line['parity_max_rel_vs_oracle_28_pairs'] = rel
        line['checksum_ok'] = bool(rel <= 1e-9)
        line['cpu_baseline'] = {'value': 28 * n / tc / 1e9, 'unit': 'Gterms/s', 'cores': 1, 'kind': 'port',
                                'sample': 'oracle distance_matrix on the first 8 profiles (28 pairs, %.2f s)' % tc}
    print(json.dumps(line), flush=True)
    ctx.free(dprof)
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='count', choices=['count', 'matrix'])
    ap.add_argument('--reads', type=int, default=100_000_000, help='reads per GPU (weak scaling)')
    ap.add_argument('--read-len', type=int, default=150)
    ap.add_argument('--k', type=int, default=12)
    ap.add_argument('--strategy', default='auto')
    ap.add_argument('--cpu-reads', type=int, default=4_000_000, help='reads in the CPU-baseline sample')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling: --reads is the TOTAL, split over the GPUs (default: weak, --reads per GPU)')
    ap.add_argument('--reduce', default='int64', choices=['int64', 'u32'],
                    help='N>1: dtype moved by the RCCL reduce (u32 halves the xGMI bytes; used only while every per-rank bin < 2^32)')
    ap.add_argument('--overlap-reduce', action='store_true',
                    help='N>1: reduce the table of step i on a second buffer while step i+1 counts')
    ap.add_argument('--profiles', type=int, default=64, help='matrix workload: number of profiles')
    ap.add_argument('--profile-reads', type=int, default=2_000_000, help='matrix workload: reads per profile')
    ap.add_argument('--metric', default='prod', choices=['prod', 'sum', 'euclidean'])
    ap.add_argument('--balance', action='store_true')
    args = ap.parse_args()

    if args.workload == 'matrix':
        return matrix_workload(args)

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
    table_ptr, bins = ctx.count_table()
    reducer = None
    if world > 1:
        reducer = kdist.TableReducer(kdist.table_as_tensor(ctx), sync=ctx.sync,
                                     balance=lambda t: ctx.balance_device(k, t.data_ptr()),
                                     mode=args.reduce, overlap=args.overlap_reduce)

    def step():
        ctx.count_begin(k, args.strategy)               # zero the 4^k table
        ctx.count_feed_device(dev_buf, nbytes)
        if reducer is not None:
            reducer.reduce_step()                       # RCCL reduce(SUM) to rank 0 (+ balance on rank 0)
        elif rank == 0:
            ctx.balance_device(k, table_ptr)            # Profile.balance on the table
        ctx.sync()

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    if reducer is not None:
        reducer.drain()
    ctx.prof_enable(True)
    ctx.prof_reset()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if reducer is not None:
        reducer.drain()
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
        ctx.d2h(out, reducer.result_ptr() if reducer is not None else table_ptr)
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
                        'frac': achieved / HBM_PEAK_GBS}
            roofline.update(pmc_traffic(dom, k, n_reads * (L + 1) * steps / launches))
            roofline.update({'avg_launch_ms': avg_ms,
                             'launches_per_step': launches / steps,
                             'algorithmic_bytes_per_launch': per_launch_bytes,
                             'pipeline_frac': alg_bytes_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'kernels_ms_per_step': {n: v[0] / steps for n, v in sorted(kern.items())},
                             # the contract prices the path against HBM; what actually limits the quad kernels since round 2
                             # (counters: profiles/r2/pmc_lds_quad.json)
                             'limiter': ('VALU + LDS issue (scatter), LDS atomics (histogram): the kernels move their bytes at 4-5 TB/s'
                                         if dom.startswith('quad') else None)})
        line = {
            'metric': 'Gbases/s k-mer counted (k=%d, %dbp synthetic)' % (k, L), 'value': value, 'unit': 'Gbases/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None, 'dtype': 'int64', 'data': 'synthetic',
            'config': {'workload': 'k=%d, %d synthetic %dbp reads per GPU resident in HBM, count%s+balance'
                                   % (k, n_reads, L, '+RCCL reduce' if world > 1 else ''),
                       'k': k, 'reads_per_gpu': n_reads, 'read_len': L, 'strategy': args.strategy,
                       'parallelism': 'reads sharded x%d, 1 reduce(%s sum)%s' % (world, args.reduce, ', overlapped with the next count' if args.overlap_reduce else '')},
            'checksum_ok': ok,
            'src_sha': source_sha(),
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
