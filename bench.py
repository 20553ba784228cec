#!/usr/bin/env python
"""bench.py -- Gbases/s of k-mer counting (k=12, 150 bp synthetic reads) on N MI355X.

Default workload (BASELINE.json metric, config 2 / 3): a "step" is one pass of the hot path over this rank's resident
batch of synthetic reads: zero the 4^k table, count every k-mer (kpal_count_feed_device), [N>1: ONE RCCL reduce of the
count tables to rank 0 -- ncclReduce(int64, sum) issued by libkpal_hip.so on its own HIP streams], balance the (merged)
table (Profile.balance).  Inputs are generated on the device before the timed region (HBM-resident); weak scaling:
every rank holds --reads reads.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R] [--k 12]

--gpus N > 1 started as a plain command spawns the N ranks itself (python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>) BEFORE anything touches the GPU, relays rank
0's JSON line and exits with the ranks' status; started under torch.distributed.run (RANK / WORLD_SIZE set) it is a rank.

The default one-GPU run appends to its JSON line, after the headline measurement, `extra`: BASELINE config 4 (k = 15,
same 100 M reads), config 5 (64-profile distance matrix at k = 12, multiset prod and euclidean), each with its own
ms_per_step / roofline / parity flag, and `end_to_end` (host-resident input: H2D, kernels, D2H of the table) for
config 2.  --no-extra skips them; --workload matrix runs config 5 alone as the headline.

Rank 0 prints ONE JSON line (DESIGN.md section 5 defines the fields).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured achievable)
FP64_VALU_PEAK_T = 39.3    # fp64 vector lane-instructions/s: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (78.6 TFLOP/s with FMA = half the guide's 157.3 TF fp32 vector rate)
FP64_MFMA_PEAK_T = 78.6    # fp64 matrix TFLOP/s (v_mfma_f64_16x16x4_f64: 2048 flop per 64 SIMD-cycles x 1024 SIMDs x 2.4 GHz)
PROFILE_ROUND = 'r6'
VALU_CYCLES_PER_INSTRUCTION = 3.44   # measured: the k = 12 scatter's opcode mix priced by tools/price_stream.py (profiles/r6/valu_prices.md)
PROFILE_DIR = os.path.join(ROOT, 'profiles', PROFILE_ROUND)


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


# ----------------------------------------------------------------------------------------------------------------------
# committed counter profiles (rocprofv3 --pmc passes of this command, tools/profile_round.sh)
# ----------------------------------------------------------------------------------------------------------------------
def _load_profile(name):
    try:
        with open(os.path.join(PROFILE_DIR, name)) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def _kernel_row(prof, kernel, k):
    """The row of `kernel` (bench's short name, e.g. quad_scatter) in a per-kernel profile."""
    base = kernel[:-len('_balanced')] if kernel.endswith('_balanced') else kernel   # (quad2_finalize_kernel<K, true, ...> is timed as quad2_finalize_balanced)
    for name, rec in prof['kernels'].items():
        if ('::%s_kernel' % base) in name and (('<%d' % k) in name or '<' not in name):
            return rec
    return None


def pmc_traffic(kernels, dom, k, input_bytes_per_step):
    """HBM bytes from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, gfx950 correction applied), scaled to
    this run's input bytes: `traffic` per launch of the dominant kernel, `traffic_step` = the sum over the kernels of a
    step.  PMC counters cannot be read from inside the bench process, so this is a PROFILE of the same command, not a
    measurement of this run: it is only reported when the profile was taken from the same kernel sources (src_sha)."""
    info = {'traffic': None, 'traffic_step': None, 'traffic_source': None, 'traffic_profile_src_sha': None}
    fname = 'pmc_hbm_traffic%s.json' % ('' if k == 12 else '_k%d' % k)
    prof = _load_profile(fname)
    if prof is None:
        info['traffic_source'] = 'no committed counter profile profiles/%s/%s' % (PROFILE_ROUND, fname)
        return info
    info['traffic_profile_src_sha'] = prof.get('src_sha')
    here = source_sha()
    if prof.get('src_sha') != here:
        info['traffic_source'] = 'profiles/%s/%s was taken from other kernel sources (src_sha %s, now %s): not reported' % (PROFILE_ROUND, fname, prof.get('src_sha'), here)
        return info
    scale = input_bytes_per_step / prof['input_bytes_per_launch_avg']
    per_kernel, total = {}, 0.0
    for name in kernels:
        row = _kernel_row(prof, name, k)
        if row is None:
            continue
        # the profile is taken on a smaller input (--reads 20 M at k = 12, 40 M at k = 15) and scaled to this run: bytes that follow the
        # input (the scatters' reads and record writes, the histogram's record reads) by the ratio of the input bytes, bytes that follow
        # the 4^k table (the histogram's merge / staging writes, the finalisation, the balance) not at all
        read = 2.0 * row.get('FETCH_SIZE', 0.0) * 1024.0     # gfx950: FETCH_SIZE x 2 (MI355X_MICROARCH.md)
        written = row.get('WRITE_SIZE', 0.0) * 1024.0
        if 'scatter' in name or name == 'quad_sample':
            per_dispatch = (read + written) * scale
        elif name.endswith('_hist'):
            per_dispatch = read * scale + written
        else:
            per_dispatch = read + written
        per_kernel[name] = per_dispatch * row['dispatches'] / max(prof['launches'], 1)
        total += per_kernel[name]
    if dom in per_kernel:
        launches_per_step = kernels[dom][1]
        info['traffic'] = per_kernel[dom] / max(launches_per_step, 1)
    info['traffic_step'] = total
    info['traffic_by_kernel_per_step'] = per_kernel
    info['traffic_source'] = ('profiles/%s/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on the same kernel sources, '
                              'scaled by input bytes)' % (PROFILE_ROUND, fname))
    return info


def pmc_limiter(dom, k):
    """What limits the dominant kernel according to the committed SQ counter pass (pmc_lds_quad.json, same kernel sources): the
    busiest of the VALU, the LDS and -- from the traffic pass -- the HBM interface."""
    prof = _load_profile('pmc_lds_quad.json' if k == 12 else 'pmc_lds_quad_k%d.json' % k)
    if prof is None or prof.get('src_sha') != source_sha():
        return None
    row = _kernel_row(prof, dom, k)
    if not row or not row.get('SQ_BUSY_CYCLES'):
        return None
    # SQ_BUSY_CYCLES is summed over the 32 shader engines: / 32 = the kernel's duration in cycles; SQ_LDS_IDX_ACTIVE is summed
    # over the 256 CUs (tools/profile_numbers.py uses the same normalisation); wave-level shares are taken against SQ_WAVE_CYCLES
    cyc = row['SQ_BUSY_CYCLES'] / 32.0
    waves = row.get('SQ_WAVE_CYCLES') or 0.0
    out = {'source': 'profiles/%s/pmc_lds_quad%s.json' % (PROFILE_ROUND, '' if k == 12 else '_k%d' % k)}
    if waves:
        out['valu_issue_share_of_wave_cycles'] = row.get('SQ_ACTIVE_INST_VALU', 0.0) / waves
        out['waiting_share_of_wave_cycles'] = row.get('SQ_WAIT_ANY', 0.0) / waves
    if row.get('SQ_LDS_IDX_ACTIVE') and cyc:
        out['lds_busy_share'] = row['SQ_LDS_IDX_ACTIVE'] / 256.0 / cyc
        out['lds_bank_conflict_share_of_lds_busy'] = row.get('SQ_LDS_BANK_CONFLICT', 0.0) / row['SQ_LDS_IDX_ACTIVE']
    # VALU: SQ_INSTS_VALU wave-instructions over the 1024 SIMDs of the chip, each at the MEASURED average issue cost of this kernel's
    # opcode mix with four waves on the SIMD (tools/valu_bench.hip + tools/price_stream.py, profiles/r6/valu_prices.md: 3.44 cycles
    # for the k = 12 scatter -- 2.5 for the plain two-operand integer / logic opcodes, 4.25 for the three-operand ones, shifts left,
    # permutes, dot products, compares, DPP and lane moves; rounds 4-5 assumed 4.0 for every instruction)
    if row.get('SQ_INSTS_VALU') and cyc:
        out['valu_busy_share'] = row['SQ_INSTS_VALU'] / 1024.0 * VALU_CYCLES_PER_INSTRUCTION / cyc
        out['valu_cycles_per_instruction'] = VALU_CYCLES_PER_INSTRUCTION
    valu = out.get('valu_busy_share')
    if valu is None:
        valu = (out.get('valu_issue_share_of_wave_cycles') or 0.0) * VALU_CYCLES_PER_INSTRUCTION
    busiest = max((('LDS', out.get('lds_busy_share') or 0.0), ('VALU issue', valu)), key=lambda t: t[1])
    out['busiest_unit'] = busiest[0]
    return out


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle port; reported next to the GPU number, never the target)
# ----------------------------------------------------------------------------------------------------------------------
def cpu_baseline(k, read_len, budget_reads, big=None):
    """CPU figures next to the GPU number (reported baselines, never the target):
    the oracle's C port (kpal/klib.py:149-170 restated) on a bounded sample of the same workload with 1
    thread and with all host cores (a private table per thread or one shared table, whichever is faster), and the pure-Python
    restatement of the reference's loop on BASELINE config 1 (the reference's own speed class).
    big: a LARGER sample of the same workload (uint8 buffer of whole reads, taken from the device buffer of the run) for the
    private-table plan, whose fixed cost -- first touch and merge of T tables of 4^k entries -- a small sample cannot amortise."""
    import oracle
    from oracle import pyref
    cores = os.cpu_count() or 1
    buf = oracle.synth_reads(2, 0, budget_reads, read_len)
    oracle.count_flat(buf[:151 * 1000], k)                      # page the table in outside the timed region
    t0 = time.perf_counter()
    c1 = oracle.count_flat(buf, k, threads=1)
    t1 = time.perf_counter() - t0
    # all cores: every thread its own 4^k table (calloc and the merge inside the timed region), as many threads as the host has
    # hardware threads and memory for (a quarter of MemAvailable at most); and ONE shared table with relaxed atomic adds, which
    # stops scaling early (cache-line ping-pong) -- a few thread counts of each are timed and the best is reported with its own
    # thread count
    avail = 64 << 30
    try:
        with open('/proc/meminfo') as fh:
            for ln in fh:
                if ln.startswith('MemAvailable:'):
                    avail = int(ln.split()[1]) * 1024
    except OSError:
        pass
    table_bytes = 8 * 4 ** k
    most_private = max(1, min(cores, 256, (avail // 4) // table_bytes))
    tn, best_threads, best_mode, tried = None, cores, 'shared', {}
    # (measured on the MI355X host, 4 M-read sample: private x 256 / 128 / 64 threads 8.0 / 4.6 / 2.2 s -- zeroing and merging T tables of
    # 128 MiB costs more than the sample's counting -- against 0.53 s for 16 threads on one shared table: one private plan is kept)
    # Round 6: the private plan is timed on `big` when the caller has one (40 M reads of the run's own device buffer: 6 Gbases against
    # ~2 s of fixed cost), every plan is compared by its RATE on its own sample, and the line says which sample the best one had.
    bases = budget_reads * read_len
    plans = [('private', min(most_private, 64), big if big is not None else buf)]
    # (128 threads with a private table each were timed in round 6 as well: 10.2 s for the 40 M reads against 7.1 s with 64 -- slower, so
    # the default line does not spend ten seconds on it again; profiles/r6/bench_k12_n1.json holds both)
    plans += [('shared', t, buf) for t in sorted(set([min(cores, 64), min(cores, 16)]), reverse=True)]
    best_rate, best_reads = 0.0, budget_reads
    for mode, threads, sample in plans:
        sample_reads = sample.size // (read_len + 1)
        t0 = time.perf_counter()
        cn = oracle.count_flat(sample, k, threads=threads, mode=mode)
        t = time.perf_counter() - t0
        tried['%s_%d%s' % (mode, threads, '' if sample is buf else '_big')] = t
        assert int(cn.sum()) == sample_reads * (read_len - k + 1)
        del cn
        rate = sample_reads * read_len / t
        if rate > best_rate:
            best_rate, tn, best_threads, best_mode, best_reads = rate, t, threads, mode, sample_reads
    assert int(c1.sum()) == budget_reads * (read_len - k + 1)
    # BASELINE config 1: 10 k reads, k = 9 through the interpreter loop
    reads1 = [bytes(r).decode() for r in oracle.synth_reads(1, 0, 10000, 150).reshape(-1, 151)[:, :150]]
    t0 = time.perf_counter()
    cp = pyref.from_sequences(reads1, 9)
    tp = time.perf_counter() - t0
    assert int(cp.sum()) == 1420000
    return {
        'value': bases / t1 / 1e9, 'unit': 'Gbases/s', 'cores': 1, 'kind': 'port',
        'sample': '%d synthetic %d bp reads, k=%d, oracle/kpal_oracle.c (1 thread: %.2f s)' % (budget_reads, read_len, k, t1),
        # the same figures as scalars (the driver's record keeps scalars only): all host cores, and the reference's own speed class
        'all_cores_value': best_rate / 1e9, 'all_cores_threads': best_threads, 'all_cores_tables': best_mode, 'all_cores_sample_reads': best_reads, 'host_cores': cores,
        # (why not more threads: every plan's seconds are in all_cores.seconds_by_threads -- on the MI355X hosts 64 and 128 threads with a
        # private 128 MiB table each count 40 M reads in 7.1 and 10.2 s, 0.85 and 0.59 Gbases/s: random increments into T x 128 MiB miss the
        # TLB and the caches on every k-mer, and one shared table with atomic adds stops scaling at ~16 threads)
        'all_cores_plans_tried': ', '.join('%s %.2f s' % kv for kv in sorted(tried.items())),
        'python_loop_value': 1.5e6 / tp / 1e9, 'python_loop_sample': 'BASELINE config 1 (10000 reads, k=9), pure-Python restatement of klib.py:149-170, 1 core',
        'all_cores': {'value': best_rate / 1e9, 'cores': best_threads, 'host_cores': cores, 'seconds': tn, 'sample_reads': best_reads,
                      'per_thread_efficiency': best_rate / (bases / t1) / best_threads,
                      'seconds_by_threads': tried,
                      'tables': best_mode,
                      'note': 'oracle/kpal_oracle.c on T threads: a private 4^k table per thread (calloc + merge timed) or one shared table with '
                              'relaxed atomic adds; best RATE of the (tables, threads) plans tried, each on its own sample (`_big`: the '
                              'private plan on %d reads taken from the run\'s device buffer -- first touch and merge of T x 4^k entries are '
                              'a fixed cost of about two seconds that a 4 M-read sample cannot amortise)' % (big.size // (read_len + 1) if big is not None else 0)},
        'python_reference_loop': {'value': 1.5e6 / tp / 1e9, 'unit': 'Gbases/s', 'cores': 1, 'seconds': tp,
                                  'sample': 'BASELINE config 1 (10000 reads, k=9) through oracle/pyref.py, the statement-by-statement restatement of kpal/klib.py:149-170'},
    }


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE config 5: the 64-profile distance matrix
# ----------------------------------------------------------------------------------------------------------------------
def matrix_profiles(ctx, k, P, profile_reads, keep_host=8):
    """P profiles (profile p = count of `profile_reads` synthetic reads, seed 100 + p) resident in HBM as int64[P][4^k]; the first
    `keep_host` also on the host for the oracle spot check."""
    import numpy as np
    n = 4 ** k
    nbytes = profile_reads * 151
    d = ctx.alloc(nbytes)
    dprof = ctx.alloc(P * n * 8)
    host = []
    for p in range(P):
        ctx.synth_reads_device(100 + p, 0, profile_reads, 150, d)
        ctx.count_begin(k)
        ctx.count_feed_device(d, nbytes)
        ctx.count_finish(to_host=False)
        ptr, _ = ctx.count_table()
        ctx.d2d(dprof + p * n * 8, ptr, n * 8)
        if p < keep_host:
            c = np.empty(n, dtype=np.int64)
            ctx.d2h(c, ptr)
            host.append(c)
    ctx.sync()
    ctx.free(d)
    return dprof, host


def matrix_measure(ctx, k, P, dprof, host, metric_name, balance, steps, warmup, check):
    """One kdistlib.distance_matrix value computation per step (kpal_distance_matrix_device) -> result dict."""
    import numpy as np
    n = 4 ** k
    metric = {'prod': 0, 'sum': 1, 'euclidean': 2}[metric_name]
    vals = None
    for _ in range(warmup):
        vals = ctx.distance_matrix_device(P, k, dprof, metric, balance)
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        vals = ctx.distance_matrix_device(P, k, dprof, metric, balance)
    ctx.sync()
    elapsed = time.perf_counter() - t0
    prof = {n_: v for n_, v in ctx.prof_get().items() if v[1] > 0}
    ctx.prof_enable(False)
    steps = max(steps, 1)
    pairs = P * (P - 1) // 2
    terms = pairs * n
    ms = elapsed / steps * 1e3
    dom = max(prof, key=lambda n_: prof[n_][0])
    dom_ms = prof[dom][0] / prof[dom][1]
    mem_bytes = 8 * P * n * (2 if balance else 1)
    if dom.startswith('gram'):
        # 10 of the 16 tile pairs of a 64-profile block are computed; 2 flop per multiply-add
        nb = (P + 63) // 64
        flops = 2.0 * n * (10 * 256 * nb + 16 * 256 * nb * (nb - 1) / 2)
        roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': flops / (dom_ms * 1e-3) / 1e12, 'peak': FP64_MFMA_PEAK_T, 'unit': 'TFLOP/s',
                    'frac': flops / (dom_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_T, 'traffic': None,
                    'memory_frac': mem_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': mem_bytes,
                    'avg_launch_ms': dom_ms}
    else:
        # fp64 lane operations a multiset term NEEDS.  matrix_rdiff / matrix_rdiff_all (prod as |1/(y+1) - 1/(x+1)| on staged reciprocals): its own
        # algebra, a subtraction and an add of the absolute value = 2.  The pair-of-counts kernels (matrix_rsum(_all) / matrix_super /
        # matrix_tile; metrics.py:118-123): |l - r|, the denominator and the division = 3.  The kernels' instruction counts are
        # higher (loader, conversions, term counts) and are not what the fraction is priced on
        slots = 2.0 if dom in ('matrix_rdiff', 'matrix_rdiff_all') else 3.0
        roofline = {'bound': 'fp64-valu', 'kernel': dom, 'achieved': terms * slots / (dom_ms * 1e-3) / 1e12, 'peak': FP64_VALU_PEAK_T,
                    'unit': 'Tinstr/s (fp64 lane operations; %.0f necessary per term)' % slots,
                    'frac': terms * slots / (dom_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_T, 'traffic': None,
                    'memory_frac': mem_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': mem_bytes,
                    'avg_launch_ms': dom_ms}
    roofline['kernels_ms_per_step'] = {n_: v[0] / steps for n_, v in sorted(prof.items())}
    out = {
        'metric': 'Gterms/s distance matrix (%d profiles k=%d, %s%s)' % (P, k, metric_name, ' balanced' if balance else ''),
        'value': terms / (elapsed / steps) / 1e9, 'unit': 'Gterms/s', 'ms_per_step': ms, 'steps': steps, 'pairs': pairs,
        'dtype': 'f64' if metric != 2 else 'int64/f64-mfma', 'roofline': roofline,
    }
    if check and host:
        import oracle
        m = len(host)
        t0 = time.perf_counter()
        want = oracle.distance_matrix_values(host, k, balance, metric_name)
        tc = time.perf_counter() - t0
        sub = np.array([vals[i * (i - 1) // 2 + j] for i in range(1, m) for j in range(i)])
        rel = float(np.max(np.abs(sub - want) / np.maximum(np.abs(want), 1e-300)))
        out['parity_max_rel_vs_oracle'] = rel
        out['parity_pairs'] = len(want)
        out['checksum_ok'] = bool(rel == 0.0) if metric == 2 else bool(rel <= 1e-9)
        out['cpu_baseline'] = {'value': len(want) * n / tc / 1e9, 'unit': 'Gterms/s', 'cores': 1, 'kind': 'port',
                               'sample': 'oracle distance_matrix on the first %d profiles (%d pairs, %.2f s)' % (m, len(want), tc)}
    return out


def matrix_workload(args):
    """BASELINE config 5 on one GPU as the headline: value = Gterms/s of the 64 x 64 lower triangle (2016 pairs x 4^12 bins)."""
    from kpal_amd import _native
    ctx = _native.Context(0)
    dprof, host = matrix_profiles(ctx, args.k, args.profiles, args.profile_reads, keep_host=0 if args.no_cpu else 8)
    r = matrix_measure(ctx, args.k, args.profiles, dprof, host, args.metric, args.balance, args.steps, args.warmup, check=not args.no_cpu)
    line = {
        'metric': r['metric'], 'value': r['value'], 'unit': r['unit'], 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': r['ms_per_step'], 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': r['dtype'], 'data': 'synthetic',
        'config': {'workload': 'BASELINE config 5: %d profiles k=%d (%d synthetic 150bp reads each, seed 100+p) resident in HBM, kdistlib.distance_matrix %s%s' % (
            args.profiles, args.k, args.profile_reads, args.metric, ' +balance' if args.balance else ''), 'pairs': r['pairs']},
        'roofline': r['roofline'], 'src_sha': source_sha(),
    }
    for key in ('parity_max_rel_vs_oracle', 'parity_pairs', 'checksum_ok', 'cpu_baseline'):
        if key in r:
            line[key] = r[key]
    print(json.dumps(line), flush=True)
    ctx.free(dprof)
    ctx.close()


# ----------------------------------------------------------------------------------------------------------------------
# counting
# ----------------------------------------------------------------------------------------------------------------------
def count_roofline(kern, k, n_bytes_step, bins, steps, ms_per_step, fused_balance):
    """Roofline object of a counting step from the per-kernel HIP-event times (`kern`: name -> (total ms, launches)).

    Algorithmic bytes: count = B_in + 8 * 4^k (every input byte read once, the table written once), balance =
    16 * 4^k.  A multi-kernel pipeline moves more; each kernel is priced on the share of those bytes that IT moves and no
    other kernel does -- the scatter that reads the input: B_in; the kernel that writes the finished table: 8 * 4^k; a
    stand-alone balance: 16 * 4^k; intermediate kernels: nothing -- and the step on all of them."""
    if not kern:
        return None
    share = {}
    finalised = 'quad2_finalize' in kern or 'quad2_finalize_balanced' in kern   # (the histogram stage then stages its forms: the finalisation writes the table)
    for name in kern:
        if name == 'quad_scatter' or name in ('chunk_scatter', 'part_scatter', 'coarse_scatter', 'count_lds_direct', 'count_global_atomic'):
            share[name] = float(n_bytes_step)
        elif name in ('quad_hist', 'chunk_hist', 'part_hist') and k <= 12 and not finalised:
            share[name] = 8.0 * bins
        elif name in ('quad2_finalize', 'quad2_finalize_balanced'):
            share[name] = 8.0 * bins
        elif name in ('balance_tiled', 'balance_inplace'):
            share[name] = 16.0 * bins
        else:
            share[name] = 0.0
    dom = max(kern, key=lambda n: kern[n][0])
    tot_ms, launches = kern[dom]
    avg_ms = tot_ms / launches
    launches_per_step = launches / steps
    per_launch = share[dom] / max(launches_per_step, 1e-9)
    achieved = per_launch / (avg_ms * 1e-3) / 1e9
    alg_step = n_bytes_step + 8.0 * bins + (0.0 if fused_balance else 16.0 * bins)
    r = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
         'avg_launch_ms': avg_ms, 'launches_per_step': launches_per_step, 'algorithmic_bytes_per_launch': per_launch,
         'algorithmic_bytes_note': 'the dominant kernel is priced on the algorithmic bytes it alone moves (scatter: the input bytes; the kernel that '
                                   'writes the finished table: 8*4^k; stand-alone balance: 16*4^k; intermediate kernels: none); the pipeline on all of them',
         'pipeline': {'algorithmic_bytes_per_step': alg_step, 'achieved': alg_step / (ms_per_step * 1e-3) / 1e9,
                      'frac': alg_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      'formula': 'B_in + 8*4^k' + ('' if fused_balance else ' + 16*4^k') + ' (count' + (' + balance fused into the finalisation' if fused_balance else ' + balance') + ')'},
         'pipeline_frac': alg_step / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
         'kernels_ms_per_step': {n: v[0] / steps for n, v in sorted(kern.items())},
         'kernel_frac_of_own_bytes': {n: (share[n] / max(v[1] / steps, 1e-9)) / (v[0] / v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS for n, v in sorted(kern.items()) if share[n] > 0}}
    per_step = {n: (v[0] / steps, v[1] / steps) for n, v in kern.items()}
    r.update(pmc_traffic(per_step, dom, k, n_bytes_step))
    if r.get('traffic_step'):
        r['traffic_step_over_algorithmic'] = r['traffic_step'] / alg_step
        r['traffic_step_rate_GBs'] = r['traffic_step'] / (ms_per_step * 1e-3) / 1e9
    if r.get('traffic'):
        # what the dominant kernel MOVES (PMC bytes per launch over its live duration) next to what it is priced on: the scatter
        # reads its input and writes as many bytes of records, so its memory pipes carry twice `achieved`
        r['traffic_rate_GBs'] = r['traffic'] / (avg_ms * 1e-3) / 1e9
        r['traffic_frac_of_peak'] = r['traffic_rate_GBs'] / HBM_PEAK_GBS
    r['limiter'] = pmc_limiter(dom, k)
    return r


def count_measure(ctx, k, dev_buf, nbytes, n_reads, read_len, strategy, steps, warmup):
    """One GPU, no communicator: `steps` timed steps of zero + count + balance over the resident buffer -> result dict."""
    import numpy as np
    ctx.count_begin(k, strategy)
    table_ptr, bins = ctx.count_table()

    def step():
        ctx.count_begin(k, strategy)                   # zero the 4^k table
        ctx.count_feed_device(dev_buf, nbytes)
        ctx.count_balance()                            # Profile.balance on the table (k >= 13: fused into its finalisation)
        ctx.sync()

    for _ in range(warmup):
        step()
    ctx.prof_enable(True)
    ctx.prof_reset()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    kern = {n: v for n, v in ctx.prof_get().items() if v[1] > 0}
    ctx.prof_enable(False)
    plan = ctx.count_last_plan()
    out = np.empty(bins, dtype=np.int64)
    ctx.d2h(out, table_ptr)
    ok = int(out.sum()) == 2 * n_reads * (read_len - k + 1)
    steps = max(steps, 1)
    ms = elapsed / steps * 1e3
    return {'ms_per_step': ms, 'value': n_reads * read_len / (elapsed / steps) / 1e9, 'checksum_ok': ok, 'plan': list(plan), 'bins': bins,
            'roofline': count_roofline(kern, k, nbytes, bins, steps, ms, fused_balance='quad2_finalize_balanced' in kern), 'table': out}


def end_to_end(ctx, k, dev_buf, nbytes, n_reads, read_len, kernel_s):
    """SURVEY.md 8d protocol: end-to-end with HOST-resident input, reported separately and never the bench value.  The reads are
    downloaded once (outside any timing), then: a plain H2D copy of them, the resident-input step (from the headline), the D2H
    of the 4^k table -- and the library's own host feed (kpal_count_feed: 64 MiB pinned staging, copies overlapped with
    counting) + balance + download, which is what a caller holding reads in host memory gets."""
    import numpy as np
    host = np.empty(nbytes, dtype=np.uint8)
    ctx.d2h(host, dev_buf)
    t0 = time.perf_counter()
    ctx.h2d(dev_buf, host)
    h2d_s = time.perf_counter() - t0
    ctx.count_begin(k)
    t0 = time.perf_counter()
    ctx.count_feed(host)
    ctx.count_balance()
    ctx.sync()
    feed_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    table = ctx.count_finish()
    d2h_s = time.perf_counter() - t0
    ok = int(table.sum()) == 2 * n_reads * (read_len - k + 1)
    bases = n_reads * read_len
    return {'h2d_s': h2d_s, 'kernel_s': kernel_s, 'd2h_s': d2h_s,
            'serial_Gbases_per_s': bases / (h2d_s + kernel_s + d2h_s) / 1e9,
            'overlapped_host_feed_s': feed_s, 'overlapped_Gbases_per_s': bases / (feed_s + d2h_s) / 1e9,
            'h2d_GBs': nbytes / h2d_s / 1e9, 'checksum_ok': ok,
            'note': 'input in pageable host memory (%d bytes); h2d_s = one plain copy; overlapped_host_feed_s = kpal_count_feed (pinned 64 MiB staging, '
                    'copy/compute overlap) + balance; never the bench value' % nbytes}


def run_extras(ctx, args, dev_buf, nbytes, headline_ms):
    """BASELINE configs 4 and 5 and the end-to-end figure, after the headline measurement (same process, same resident buffer)."""
    import numpy as np
    extra = {}
    L = args.read_len

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            extra[name] = fn()
        except Exception as e:   # an extra never takes the headline line down
            extra[name] = {'error': '%s: %s' % (type(e).__name__, e)}
        extra[name]['wall_s'] = time.perf_counter() - t0

    guarded('end_to_end', lambda: end_to_end(ctx, args.k, dev_buf, nbytes, args.reads, L, headline_ms * 1e-3))

    def skew():
        """Not every read set is i.i.d. uniform: 1 GiB of (a) the uniform reads, (b) the same with 2 % of the reads replaced by poly-A /
        (AC)n reads, (c) a homopolymer, counted at k = 12 and k = 15 through AUTO, whole buffer resident in HBM (kernel time of
        begin + feed + finish, best of three).  Parity of exactly these inputs at exactly this size: tests/test_gpu_skew_full.py."""
        n_reads = min((1 << 30) // (L + 1), args.reads)     # (1 GiB, or the whole resident buffer when that is smaller)
        n = n_reads * (L + 1)
        host = np.empty(n, dtype=np.uint8)
        ctx.synth_reads_device(2, 0, n_reads, L, dev_buf)
        ctx.d2h(host, dev_buf)
        rs = np.random.RandomState(5)
        low = host.reshape(n_reads, L + 1).copy()
        hit = np.flatnonzero(rs.rand(n_reads) < 0.02)
        poly = rs.rand(hit.size) < 0.5
        low[hit[poly], :L] = ord('A')
        low[hit[~poly], :L] = np.frombuffer((b'AC' * L)[:L], dtype=np.uint8)
        cases = (('uniform', host), ('low_complexity_2pct', low.reshape(-1)), ('homopolymer', np.full(n, ord('A'), dtype=np.uint8)))
        out = {}
        for k2 in (12, 15):
            for name, buf in cases:
                ctx.h2d(dev_buf, buf)
                best, plan = None, None
                for it in range(4):
                    ctx.prof_enable(True)
                    ctx.prof_reset()
                    ctx.count_begin(k2)
                    ctx.count_feed_device(dev_buf, n)
                    plan = ctx.count_last_plan()
                    ctx.count_finish(to_host=False)
                    ctx.sync()
                    ms = sum(v[0] for v in ctx.prof_get().values())
                    ctx.prof_enable(False)
                    if it and (best is None or ms < best):     # (the first pass is the warm-up)
                        best = ms
                out['k%d_%s' % (k2, name)] = {'ms': best, 'Gbases_per_s': n_reads * L / best / 1e6, 'plan': '%s/%d/%d' % plan}
        return out
    guarded('skew', skew)

    def k15():
        ctx.synth_reads_device(4, 0, args.reads, L, dev_buf)             # seed 4
        r = count_measure(ctx, 15, dev_buf, nbytes, args.reads, L, 'auto', 3, 1)
        r.pop('table')
        r.update({'metric': 'Gbases/s k-mer counted (k=15, %dbp synthetic)' % L, 'unit': 'Gbases/s', 'steps': 3, 'warmup': 1, 'dtype': 'int64',
                  'config': 'BASELINE config 4: k=15 (8 GiB table), %d reads resident in HBM, count+balance' % args.reads})
        return r
    guarded('k15', k15)

    def matrices():
        dprof, host = matrix_profiles(ctx, 12, args.profiles, args.profile_reads, keep_host=0 if args.no_cpu else 8)
        try:
            extra['matrix_prod'] = matrix_measure(ctx, 12, args.profiles, dprof, host, 'prod', False, 3, 1, check=not args.no_cpu)
            extra['matrix_sum'] = matrix_measure(ctx, 12, args.profiles, dprof, host, 'sum', False, 3, 1, check=not args.no_cpu)
            extra['matrix_euclidean'] = matrix_measure(ctx, 12, args.profiles, dprof, host, 'euclidean', False, 3, 1, check=not args.no_cpu)
        finally:
            ctx.free(dprof)
        return {'config': 'BASELINE config 5: %d profiles k=12 (%d reads each, seed 100+p) resident in HBM; see matrix_prod / matrix_sum / matrix_euclidean' % (args.profiles, args.profile_reads)}
    guarded('matrix', matrices)

    def fasta_end_to_end():
        """Profile.from_fasta on a FILE (what `kpal count` does, kmer.py:112-146 -> klib.py:97-112): a 3.9 GB FASTA in 60-column
        lines (records of ~100 Mbases, bases from the 8d generator) in tmpfs, read by the library itself -- page cache ->
        pinned staging -> HBM -> flattened -> counted, pipelined; + the download of the table.  tools/clibench.py is the
        full version (8 GB, the CLI entry, shards)."""
        import tempfile
        from kpal_amd import klib
        width, lines = 60, 64_000_000
        d = '/dev/shm' if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK) else tempfile.gettempdir()
        path = os.path.join(d, 'kpal_bench_%d.fa' % os.getpid())
        host = np.empty(lines * (width + 1), dtype=np.uint8)
        ctx.synth_reads_device(77, 0, lines, width, dev_buf)
        ctx.d2h(host, dev_buf)
        try:
            with open(path, 'wb') as fh:
                per = 1_600_000
                for r, at in enumerate(range(0, lines, per)):
                    fh.write(b'>chr%d synthetic\n' % (r + 1))
                    fh.write(host[at * (width + 1):min(at + per, lines) * (width + 1)].data)
            del host
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                with open(path) as fh:
                    p = klib.Profile.from_fasta(fh, args.k)
                    p.counts                                      # (the download of the table belongs to the figure: since round 6 it happens on first access)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            per_record = per * width - args.k + 1
            records = (lines + per - 1) // per
            want = (records - 1) * per_record + ((lines - (records - 1) * per) * width - args.k + 1)
            return {'seconds': best, 'Gbases_per_s': lines * width / best / 1e9, 'file_GBs': os.path.getsize(path) / best / 1e9,
                    'file_bytes': os.path.getsize(path), 'checksum_ok': int(p.total) == want, 'dir': d,
                    'note': 'Profile.from_fasta(open(path), k) incl. the D2H of the table; input in the page cache, never the bench value'}
        finally:
            if os.path.exists(path):
                os.unlink(path)
    if nbytes >= 64_000_000 * 61:
        guarded('fasta_end_to_end', fasta_end_to_end)
    return extra


# ----------------------------------------------------------------------------------------------------------------------
# launcher, rank supervisors, workers
#
#   python bench.py --gpus N                   plain command: launch_ranks() starts torch.distributed.run (never touches a GPU)
#     -> torch.distributed.run -> N ranks     each rank is a SUPERVISOR (supervise_rank): it never touches a GPU either, starts
#       -> bench.py --worker --attempt 1      the worker that does the GPU work, and watches it: no start / no end within the
#       -> bench.py --worker --attempt 2      limits, a non-zero exit, or a peer's failure flag -> the worker's process group is
#                                              killed and a FRESH worker runs the conservative mode (torch.distributed reduce,
#                                              serial).  The driver starts the ranks through torch.distributed.run itself: the
#                                              same supervisors run there, so a hang of the first multi-GPU run still ends in a line.
# Supervisors of one job share a directory of flag files (KPAL_BENCH_RUN_DIR, or one named after MASTER_PORT and the common
# parent process): a<attempt>.started.<rank>, a<attempt>.failed.<rank>, a<attempt>.line (rank 0's JSON line as soon as the
# headline mode is measured and verified: it survives a hang in a later mode), a<attempt>.done.
# ----------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def kill_group(proc):
    """End the process group a child was started in (start_new_session=True); never a pattern kill."""
    import signal
    for sig in (signal.SIGTERM, signal.SIGKILL):
        if proc.poll() is not None:
            return
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            return
        try:
            proc.wait(timeout=5)
        except subprocess.TimeoutExpired:
            pass


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) as a plain command: start the N ranks as a fresh child process group BEFORE this
    process touches the GPU (it never does), relay rank 0's JSON line, exit with the child's status.  A wall-clock limit
    (KPAL_BENCH_LAUNCH_TIMEOUT, default 2400 s: both attempts of the rank supervisors fit) ends a job that never returns."""
    import tempfile
    run_dir = tempfile.mkdtemp(prefix='kpal_bench_')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, KPAL_BENCH_RUN_DIR=run_dir)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # the host driver only supports dmabuf IPC (RCCL across processes)
    limit = float(os.environ.get('KPAL_BENCH_LAUNCH_TIMEOUT', '2400'))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
    last = [None]

    def relay():
        for raw in p.stdout:
            text = raw.decode('utf-8', 'replace').rstrip('\n')
            try:
                obj = json.loads(text)
                if isinstance(obj, dict) and 'metric' in obj:
                    last[0] = text
                    continue
            except ValueError:
                pass
            print(text, file=sys.stderr, flush=True)

    import threading
    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    try:
        rc = p.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print('bench.py: the ranks did not finish within %.0f s: killing them' % limit, file=sys.stderr, flush=True)
        kill_group(p)
        rc = 124
    reader.join(timeout=10)
    import shutil
    shutil.rmtree(run_dir, ignore_errors=True)
    if last[0] is not None:
        print(last[0], flush=True)
    sys.exit(rc if rc else (0 if last[0] is not None else 1))


def bench_run_dir():
    """Directory of the flag files the supervisors (and workers) of one job share."""
    import tempfile
    d = os.environ.get('KPAL_BENCH_RUN_DIR')
    if not d:   # started by somebody else's torch.distributed.run (the driver): all ranks have the same parent and port
        d = os.path.join(tempfile.gettempdir(), 'kpal_bench_%s_%d' % (os.environ.get('MASTER_PORT', '0'), os.getppid()))
    os.makedirs(d, exist_ok=True)
    return d


def flag_path(attempt, what, rank=None):
    return os.path.join(bench_run_dir(), 'a%d.%s%s' % (attempt, what, '' if rank is None else '.%d' % rank))


def set_flag(attempt, what, rank=None, text=''):
    path = flag_path(attempt, what, rank)
    tmp = '%s.tmp%d' % (path, os.getpid())
    with open(tmp, 'w') as fh:
        fh.write(text)
    os.replace(tmp, path)      # (atomic: a reader sees the whole text or no file)


def supervise_rank(args):
    """One rank of an N > 1 job, under torch.distributed.run: starts the worker process and watches it (see the head of this
    section).  Never imports torch, never touches the GPU.  Exit status: 0 iff a verified line was printed."""
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    run_dir = bench_run_dir()
    startup_limit = float(os.environ.get('KPAL_BENCH_STARTUP_TIMEOUT', '420'))   # a cold `import torch` takes 1-2 min on a fresh box
    run_limit = float(os.environ.get('KPAL_BENCH_RUN_TIMEOUT', '420'))           # generate + warm-up + every mode + verification: under a minute; rank 0's CPU baseline sample: ~1 min
    reason = None
    for attempt in (1, 2):
        argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ['--worker', '--attempt', str(attempt)]
        if attempt == 2:
            argv += ['--reduce-via', 'torch', '--serial-reduce', '--fallback-reason', reason or 'attempt 1 failed']
        child = subprocess.Popen(argv, env=dict(os.environ, KPAL_BENCH_RUN_DIR=run_dir), start_new_session=True)
        t0 = time.time()
        started_at, why = None, None
        while True:
            rc = child.poll()
            if rc is not None:
                why = None if rc == 0 else 'the worker of rank %d exited with status %d' % (rank, rc)
                break
            now = time.time()
            if started_at is None and os.path.exists(flag_path(attempt, 'started', rank)):
                started_at = now
            peers = [r for r in range(world) if r != rank and os.path.exists(flag_path(attempt, 'failed', r))]
            if peers:
                try:
                    with open(flag_path(attempt, 'failed', peers[0])) as fh:
                        told = fh.read().strip()
                except OSError:
                    told = ''
                why = 'rank %d gave up on attempt %d (%s)' % (peers[0], attempt, told or 'no reason given')
            elif started_at is None and now - t0 > startup_limit:
                why = 'the worker of rank %d did not start within %.0f s' % (rank, startup_limit)
            elif started_at is not None and now - started_at > run_limit:
                why = 'the worker of rank %d did not finish within %.0f s' % (rank, run_limit)
            if why:
                break
            time.sleep(0.2)
        if why is None:
            return 0
        # this attempt is over for every rank: say so, end the worker, keep what was measured
        if not os.path.exists(flag_path(attempt, 'failed', rank)):
            set_flag(attempt, 'failed', rank, why)
        kill_group(child)
        print('bench.py supervisor (rank %d, attempt %d): %s' % (rank, attempt, why), file=sys.stderr, flush=True)
        if os.path.exists(flag_path(attempt, 'done')):
            return 0                                   # rank 0 had printed its line: only the teardown failed
        line_file = flag_path(attempt, 'line')
        if os.path.exists(line_file):
            if rank == 0:                              # the headline mode was measured and verified before the failure
                with open(line_file) as fh:
                    line = json.loads(fh.read())
                line['later_modes_failed'] = why
                print(json.dumps(line), flush=True)
                set_flag(attempt, 'done')
            return 0
        reason = reason or why
    return 1


def join_process_group(backend, rank, world, attempt, device_id=None):
    """The workers' process group.  Under torch.distributed.run the agent's store is used through a per-attempt prefix (keys of
    a killed first attempt -- its ncclUniqueId -- must not be read by the second); without an agent rank 0's worker hosts the
    store, on MASTER_PORT + attempt - 1."""
    import datetime
    import torch.distributed as td
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ['MASTER_PORT'])
    timeout = datetime.timedelta(minutes=15)
    if os.environ.get('TORCHELASTIC_USE_AGENT_STORE', '') == 'True':
        store = td.PrefixStore('kpal_bench/a%d' % attempt, td.TCPStore(addr, port, world, False, timeout))
    else:
        store = td.TCPStore(addr, port + attempt - 1, world, rank == 0, timeout)
    kw = {'device_id': device_id} if device_id is not None else {}
    td.init_process_group(backend, store=store, rank=rank, world_size=world, timeout=timeout, **kw)


def stub_rank(args):
    """--stub: the multi-process plumbing alone, on CPU (gloo) and WITHOUT any counting -- supervisor, rendezvous, shard
    arithmetic, one reduce(SUM) of int64 tables to rank 0 per step, max-over-ranks timing, the per-bin comparison of the merged
    table with the single-stream one, rank 0's JSON line.  Exists so that the launcher / supervisor path of `python bench.py
    --gpus N` can be tested where there is no GPU; `value` is null.  --stub-hang-attempt A: every worker of attempt A stops
    responding after it has started; --stub-fail-rank R: rank R's worker of attempt 1 dies before the rendezvous;
    --stub-hang-after-headline: rank 0 publishes its line (as after the headline reduce mode), then every worker hangs."""
    import torch
    import torch.distributed as td
    from kpal_amd import dist as kdist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    # what RCCL chose (rings / trees, channels, transports, protocol per collective) goes to one file per rank; rank 0 condenses
    # its own into one string of the line (rccl_info)
    nccl_log = None
    if os.environ.get('KPAL_BENCH_RUN_DIR') and os.environ.get('KPAL_BENCH_NCCL_INFO', '1') != '0':
        nccl_log = os.path.join(os.environ['KPAL_BENCH_RUN_DIR'], 'nccl_attempt%d_rank%d.log' % (args.attempt, rank))
        os.environ.setdefault('NCCL_DEBUG', 'INFO')
        os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH,TUNING')
        os.environ.setdefault('NCCL_DEBUG_FILE', nccl_log)
    if args.worker:
        set_flag(args.attempt, 'started', rank)
    if args.attempt == 1 and args.stub_fail_rank == rank:
        sys.exit(3)
    if args.stub_hang_attempt == args.attempt:
        time.sleep(3600)
    join_process_group('gloo', rank, world, args.attempt)
    total = 1000 * world + 7
    first, n = kdist.shard_range(total, rank, world)
    table = torch.zeros(4 ** 4, dtype=torch.int64)
    idx = torch.arange(first, first + n) % table.numel()
    table.index_add_(0, idx, torch.ones(n, dtype=torch.int64))          # "counts" of this rank's shard of unit ids
    td.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t = table.clone()
        kdist.reduce_counts(t, dst=0)
    td.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    td.all_reduce(el, op=td.ReduceOp.MAX)
    if rank == 0:
        want = torch.zeros_like(table)
        want.index_add_(0, torch.arange(total) % table.numel(), torch.ones(total, dtype=torch.int64))
        same = bool(torch.equal(t, want))
        line = {'metric': 'stub (plumbing only, no GPU work)', 'value': None, 'unit': 'Gbases/s', 'n_gpus': world, 'rccl_ranks': td.get_world_size(),
                'backend': 'gloo', 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': float(el.item()) / max(args.steps, 1) * 1e3,
                'checksum_ok': same, 'merged_equals_single_stream': same, 'attempt': args.attempt, 'scaling': 'weak', 'data': 'synthetic'}
        if args.fallback_reason:
            line['fallback_reason'] = args.fallback_reason
        if args.worker and args.stub_hang_after_headline:
            set_flag(args.attempt, 'line', text=json.dumps(line))     # the headline mode is measured and verified ...
    if args.stub_hang_after_headline:
        time.sleep(3600)                                              # ... and a later mode never returns
    if rank == 0:
        print(json.dumps(line), flush=True)
        if args.worker:
            set_flag(args.attempt, 'done')
    td.barrier()
    td.destroy_process_group()


def rccl_info(path):
    """One short string from a rank's NCCL_DEBUG=INFO log: version, channels, rings / trees, transports seen, and the algorithm /
    protocol lines of the tuning subsystem if it printed any.  Never raises: '' when there is nothing to say."""
    try:
        import re
        with open(path, errors='replace') as fh:
            text = fh.read()
        out = []
        m = re.search(r'(RCCL version [^\n]*|NCCL version [^\n]*)', text)
        if m:
            out.append(m.group(1).strip()[:60])
        chans = re.findall(r'Channel (\d+)/(\d+)', text)
        if chans:
            out.append('%s channels' % chans[-1][1])
        for word, label in (('Connected all rings', 'rings'), ('Connected all trees', 'trees')):
            if word in text:
                out.append(label)
        via = sorted(set(re.findall(r'via ([A-Za-z0-9/_]+)', text)))
        if via:
            out.append('via ' + '+'.join(via[:4]))
        algo = sorted(set(re.findall(r'(?:Algo(?:rithm)?\s*[=:]?\s*)(\w+)[^\n]{0,40}?(?:proto(?:col)?\s*[=:]?\s*)(\w+)', text)))
        if algo:
            out.append('algo/proto ' + ','.join('%s/%s' % a for a in algo[:4]))
        return '; '.join(out)[:240]
    except Exception:
        return ''


def multi_gpu_worker(args):
    """One worker of an N > 1 job (BASELINE config 3): this rank's shard resident in HBM; per step zero + count + ONE reduce of
    the 4^k tables to rank 0 + balance there.  Up to three reduce modes are measured in one invocation, each with its own
    warm-up and EXACTLY --steps timed steps between barriers, max over the ranks: the in-library RCCL reduce pipelined with
    the next count (the design default), the same serially, and torch.distributed's reduce serially.  After each mode rank 0
    compares the merged + balanced table of its last step BIN FOR BIN with the single-stream count of all shards (recounted
    on rank 0 into one table).  The headline is the first mode, in that order, whose table is right; the others are `extra`."""
    import numpy as np
    import torch
    import torch.distributed as td
    from kpal_amd import _native, dist as kdist

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU (no CPU fallback for the hot path)')
    # KPAL_BENCH_SHARED_GPU=1 (tests on a one-GPU box): every rank works on device 0 and the workers' process group is gloo -- RCCL
    # refuses two ranks on one device, so only the torch.distributed reducer is measured; what this mode is for is the N > 1 code
    # around it (supervisors with real GPU workers, shards, the per-bin check of the merged table) running with real counting
    shared_gpu = os.environ.get('KPAL_BENCH_SHARED_GPU', '') == '1'
    if shared_gpu:
        local_rank = 0
    backend = 'gloo' if shared_gpu else 'nccl'
    torch.cuda.set_device(local_rank)
    ctx = _native.Context(local_rank)
    # what RCCL chose (rings / trees, channels, transports, protocol per collective) goes to one file per rank; rank 0 condenses
    # its own into one string of the line (rccl_info)
    nccl_log = None
    if os.environ.get('KPAL_BENCH_RUN_DIR') and os.environ.get('KPAL_BENCH_NCCL_INFO', '1') != '0':
        nccl_log = os.path.join(os.environ['KPAL_BENCH_RUN_DIR'], 'nccl_attempt%d_rank%d.log' % (args.attempt, rank))
        os.environ.setdefault('NCCL_DEBUG', 'INFO')
        os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH,TUNING')
        os.environ.setdefault('NCCL_DEBUG_FILE', nccl_log)
    if args.worker:
        set_flag(args.attempt, 'started', rank)
        join_process_group(backend, rank, world, args.attempt, device_id=None if shared_gpu else torch.device('cuda', local_rank))
    else:   # a rank started without the supervisor (KPAL_BENCH_NO_SUPERVISOR=1)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        kw = {} if shared_gpu else {'device_id': torch.device('cuda', local_rank)}
        td.init_process_group(backend, rank=rank, world_size=world, **kw)

    k, L = args.k, args.read_len
    seed = 3
    if args.strong:
        shards = [kdist.shard_range(args.reads, r, world) for r in range(world)]   # fixed total, contiguous shards
    else:
        shards = [(r * args.reads, args.reads) for r in range(world)]             # shard s = reads [s*R, (s+1)*R)
    first_read, n_reads = shards[rank]
    total_reads = sum(n for _, n in shards)
    nbytes = n_reads * (L + 1)
    dev_buf = ctx.alloc(max(n for _, n in shards) * (L + 1))
    ctx.synth_reads_device(seed, first_read, n_reads, L, dev_buf)
    ctx.sync()
    ctx.count_begin(k, args.strategy)                   # allocates the table once
    table_ptr, bins = ctx.count_table()

    # ---- which reduce modes ------------------------------------------------------------------------------------------
    library_error = None
    library = args.reduce_via == 'library'
    if shared_gpu and library and not os.environ.get('KPAL_RCCL_LIBRARY'):
        # (with KPAL_RCCL_LIBRARY the tests put a stand-in behind the library's communicator: tests/native/fake_rccl.cpp)
        library = False
        library_error = 'KPAL_BENCH_SHARED_GPU=1: the ranks share one device, which RCCL refuses'
    if library:
        # Every rank first proves that it can bind RCCL -- rank 0 by creating the id, the others by loading the library alone
        # (no id, no bootstrap listener): ncclCommInitRank is a collective, a rank that failed before it would leave the
        # others waiting.  Any failure -> all ranks take the torch.distributed reducer and the line says so.
        my_id = None
        try:
            if rank == 0:
                my_id = _native.comm_unique_id()
            else:
                _native.comm_probe()
        except Exception as e:
            library_error = '%s: %s' % (type(e).__name__, e)
        able = torch.tensor([0 if library_error else 1], dtype=torch.int32, device='cuda')
        td.all_reduce(able, op=td.ReduceOp.MIN)
        if int(able.item()) == 0:
            library = False
            library_error = library_error or 'another rank could not bind RCCL'
    if library:
        # the communicator of the library: rank 0's id reaches the others through the workers' process group
        ident = torch.zeros(_native.COMM_ID_BYTES, dtype=torch.uint8, device='cuda')
        if rank == 0:
            ident.copy_(torch.frombuffer(bytearray(my_id), dtype=torch.uint8))
        td.broadcast(ident, src=0)
        torch.cuda.synchronize()
        # ncclCommInitRank is a collective over a bootstrap socket: time-boxed on its own (a rank that cannot reach the others must
        # not eat the whole run limit) -- a timeout ends this attempt with its reason, the supervisors start the conservative one
        import threading
        init_error, init_done = [None], threading.Event()
        comm_id = bytes(ident.cpu().numpy().tobytes())

        def comm_init():
            try:
                ctx.comm_init(rank, world, comm_id)
            except Exception as e:
                init_error[0] = '%s: %s' % (type(e).__name__, e)
            finally:
                init_done.set()
        threading.Thread(target=comm_init, daemon=True).start()
        init_limit = float(os.environ.get('KPAL_BENCH_COMM_INIT_TIMEOUT', '120'))
        if not init_done.wait(init_limit):
            init_error[0] = 'kpal_comm_init (ncclCommInitRank of the library) did not return within %.0f s on rank %d' % (init_limit, rank)
        if init_error[0]:
            why = 'library_rccl_error: ' + init_error[0]
            print('bench.py worker: ' + why, file=sys.stderr, flush=True)
            if args.worker:
                set_flag(args.attempt, 'failed', rank, why)
            os._exit(3)                                 # (a thread may still sit in the collective: no orderly teardown)
    modes = []
    if library:
        modes = ['library_serial', 'library_pipelined'] if args.serial_reduce else ['library_pipelined', 'library_serial']
    modes.append('torch_serial')
    # the bin-range merge (ncclReduceScatter + mirrored-range exchange: every rank keeps and balances 1 / W of the table): the
    # merge k >= 13 needs (8 GiB per rank at k = 15); measured at every k as `extra` -- LAST, a collective pattern no run has
    # exercised yet must not stand between the run and its other figures -- the headline only when asked (--range-merge)
    if library and kdist.range_merge_supported(k, world):   # (RangeIndex::valid(): stricter than 4^k >= world^2 for odd log2(world))
        if args.range_merge:
            modes.insert(0, 'library_range')
        else:
            modes.append('library_range')
    if args.only_headline_mode:
        modes = modes[:1]

    def fence():
        ctx.sync()                                      # the context's streams (incl. a pipelined reduce)
        torch.cuda.synchronize()
        td.barrier()
        torch.cuda.synchronize()

    def measure(mode):
        reducer = None
        if mode == 'torch_serial':
            reducer = kdist.TableReducer(kdist.table_as_tensor(ctx), sync=ctx.sync, balance=lambda t: ctx.balance_device(k, t.data_ptr()),
                                         mode=args.reduce, overlap=args.overlap_reduce)
        pipelined = mode == 'library_pipelined'
        ranged = mode == 'library_range'

        def step():
            ctx.count_begin(k, args.strategy)           # zero the 4^k table
            ctx.count_feed_device(dev_buf, nbytes)
            if ranged:
                # ONE ncclReduceScatter (every rank keeps its range of the merged table) + the balance of that range through one
                # all-to-all of the mirrored entries: queued by the library on the context's stream
                ctx.comm_reduce_scatter_table(balance=True)
                ctx.sync()
            elif reducer is None:
                # ONE ncclReduce(int64, sum) to rank 0 + balance there, queued by the library: no host synchronisation in the step at
                # k <= 12 (k >= 13: the first feed of a count reads one word back, include/kpal_hip.h: kpal_count_feed_device)
                ctx.comm_reduce_table(0, balance=True, pipelined=pipelined)
                if not pipelined:
                    ctx.sync()
            else:
                reducer.reduce_step()                   # torch.distributed.reduce(SUM) to rank 0 (+ balance on rank 0)
                ctx.sync()

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
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        td.all_reduce(t, op=td.ReduceOp.MAX)
        if ranged:
            ctx.comm_gather_table()                     # (collective, outside the timed region: the whole vector on rank 0 for the check)
            ctx.sync()
        merged = None
        if rank == 0:
            merged = np.empty(bins, dtype=np.int64)
            ctx.d2h(merged, ctx.comm_merged_table()[0] if reducer is None else reducer.result_ptr())
        return {'elapsed': float(t.item()), 'prof': {n: v for n, v in prof.items() if v[1] > 0}, 'merged': merged}

    want = [None]

    def single_stream():
        """Rank 0: every shard counted into ONE table, one after the other, then Profile.balance -- what the merged table
        must equal ('equals the single-stream count of all reads')."""
        if want[0] is None:
            ctx.count_begin(k, args.strategy)
            for f, n in shards:
                ctx.synth_reads_device(seed, f, n, L, dev_buf)
                ctx.count_feed_device(dev_buf, n * (L + 1))
            ctx.count_balance()
            want[0] = ctx.count_finish()
            ctx.synth_reads_device(seed, first_read, n_reads, L, dev_buf)      # this rank's own shard again
            ctx.sync()
        return want[0]

    def make_line(mode, results):
        r = results[mode]
        steps = max(args.steps, 1)
        ms = r['elapsed'] / steps * 1e3
        lib = mode.startswith('library')
        line = {
            'metric': 'Gbases/s k-mer counted (k=%d, %dbp synthetic)' % (k, L), 'value': total_reads * L / (r['elapsed'] / steps) / 1e9, 'unit': 'Gbases/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms,
            'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None, 'dtype': 'int64', 'data': 'synthetic',
            'config': {'workload': 'k=%d, %d synthetic %dbp reads per GPU resident in HBM, count+RCCL reduce+balance' % (k, n_reads, L),
                       'k': k, 'reads_per_gpu': n_reads, 'read_len': L, 'strategy': args.strategy,
                       'parallelism': ('reads sharded x%d, 1 ncclReduceScatter(int64 sum) of the 4^k table + 1 all-to-all of 4^k/W mirrored entries per step' if mode == 'library_range'
                                       else 'reads sharded x%d, 1 ncclReduce(int64 sum) of the 4^k table to rank 0 per step') % world,
                       'reduce_mode': mode, 'attempt': args.attempt, 'merged_equals_single_stream': r['same']},
            'checksum_ok': r['sum_ok'], 'merged_equals_single_stream': r['same'], 'rccl_ranks': td.get_world_size(),
            'reduce_via': 'library' if lib else 'torch', 'pipelined_reduce': mode == 'library_pipelined', 'reduce_mode': mode, 'attempt': args.attempt,
            'src_sha': source_sha(), 'roofline': count_roofline(r['prof'], k, nbytes, bins, steps, ms, fused_balance=False),
        }
        extra = {}
        for other, o in results.items():
            key = {'library_pipelined': 'pipelined_reduce', 'library_serial': 'serial_reduce', 'torch_serial': 'torch_reduce', 'library_range': 'range_merge'}[other]
            extra[key] = {'ms_per_step': o['elapsed'] / steps * 1e3, 'value': total_reads * L / (o['elapsed'] / steps) / 1e9,
                          'merged_equals_single_stream': o['same'], 'checksum_ok': o['sum_ok'],
                          'kernels_ms_per_step': {n: v[0] / steps for n, v in sorted(o['prof'].items())}}
            line['config'][key + '_ms_per_step'] = extra[key]['ms_per_step']       # (scalars: the driver's record keeps them)
        line['extra'] = extra
        if shared_gpu:
            line['config']['shared_gpu'] = True
            line['rccl_ranks'] = 0
        if nccl_log:
            line['config']['rccl_info'] = rccl_info(nccl_log)
        if cpu[0] is not None:
            line['cpu_baseline'] = cpu[0]
        if library_error:
            line['library_rccl_error'] = library_error
        if args.fallback_reason:
            line['fallback_reason'] = args.fallback_reason
            line['config']['fallback_reason'] = args.fallback_reason[:100]
        return line

    results, headline = {}, None
    cpu = [None]
    for mode in modes:
        r = measure(mode)
        if rank == 0:
            if cpu[0] is None and not args.no_cpu:
                # the same bounded CPU sample as at N = 1, on rank 0's host cores, outside every timed region (the other ranks wait in
                # the next mode's first collective, or in the final fence)
                try:
                    cpu[0] = cpu_baseline(k, L, args.cpu_reads)
                except Exception as e:       # never lose the line over the reported baseline
                    cpu[0] = {'error': '%s: %s' % (type(e).__name__, e)}
            w = single_stream()
            r['same'] = bool(np.array_equal(r.pop('merged'), w))
            r['sum_ok'] = int(w.sum()) == 2 * total_reads * (L - k + 1)
            results[mode] = r
            if headline is None and r['same'] and r['sum_ok']:
                headline = mode
            if headline is not None and args.worker:
                set_flag(args.attempt, 'line', text=json.dumps(make_line(headline, results)))
    ok = True
    if rank == 0:
        ok = headline is not None
        print(json.dumps(make_line(headline or modes[0], results)), flush=True)
        if args.worker:
            set_flag(args.attempt, 'done')
    fence()
    ctx.free(dev_buf)
    ctx.close()
    td.destroy_process_group()
    if not ok:
        sys.exit('the merged table differs from the single-stream count')


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
    ap.add_argument('--cpu-big-reads', type=int, default=40_000_000, help='reads in the sample of the private-table all-cores plan (N = 1; taken from the device buffer)')
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='N=1: skip BASELINE configs 4 / 5 and the end-to-end figures after the headline')
    ap.add_argument('--strong', action='store_true',
                    help='strong scaling: --reads is the TOTAL, split over the GPUs (default: weak, --reads per GPU)')
    ap.add_argument('--reduce-via', default='library', choices=['library', 'torch'],
                    help='N>1: who issues the RCCL reduce of the headline -- libkpal_hip.so on its own streams (default; the torch.distributed '
                         'reduce is measured as well, as `extra`), or torch.distributed only (kpal_amd.dist.TableReducer)')
    ap.add_argument('--serial-reduce', action='store_true',
                    help='N>1, library: the headline is count -> reduce -> balance in ONE stream per step instead of the pipelined default (the reduce + '
                         'balance of step i run on a copy of the table and a second stream while step i+1 counts); the other form is measured as `extra`')
    ap.add_argument('--only-headline-mode', action='store_true', help='N>1: measure the headline reduce mode only')
    ap.add_argument('--range-merge', action='store_true',
                    help='N>1, library: the headline is the bin-range merge (ncclReduceScatter + mirrored-range exchange: every rank keeps and balances '
                         '1 / W of the table) instead of the reduce to rank 0; it is measured as `extra` either way')
    ap.add_argument('--reduce', default='int64', choices=['int64', 'u32'], help='N>1, torch reduce only: dtype moved by the reduce')
    ap.add_argument('--overlap-reduce', action='store_true', help='N>1, torch reduce only: reduce on a second buffer while the next step counts')
    ap.add_argument('--stub', action='store_true', help='plumbing self-test on CPU/gloo, no counting (tests of the launcher and the supervisors)')
    ap.add_argument('--stub-hang-attempt', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--stub-fail-rank', type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument('--stub-hang-after-headline', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)           # set by supervise_rank
    ap.add_argument('--attempt', type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument('--fallback-reason', default=None, help=argparse.SUPPRESS)
    ap.add_argument('--profiles', type=int, default=64, help='matrix workload: number of profiles')
    ap.add_argument('--profile-reads', type=int, default=2_000_000, help='matrix workload: reads per profile')
    ap.add_argument('--metric', default='prod', choices=['prod', 'sum', 'euclidean'])
    ap.add_argument('--balance', action='store_true')
    args = ap.parse_args()

    # ---- launcher and supervisors: nothing above or in here imports torch.cuda, the native library or anything else that opens the GPU
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return launch_ranks(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # KPAL_BENCH_FORCE_MULTI=1 (tests on a one-GPU box): a world of ONE rank takes the multi-GPU code path -- supervisor, worker,
    # communicators of size one, every reduce mode, the per-bin comparison with the single-stream count
    multi = world > 1 or (os.environ.get('KPAL_BENCH_FORCE_MULTI', '') == '1' and 'RANK' in os.environ)
    if multi and not args.worker and os.environ.get('KPAL_BENCH_NO_SUPERVISOR', '') != '1':
        sys.exit(supervise_rank(args))
    if args.stub:
        return stub_rank(args)
    if args.workload == 'matrix':
        return matrix_workload(args)
    args.gpus = world
    if multi:
        return multi_gpu_worker(args)

    import numpy as np   # noqa: F401
    from kpal_amd import _native

    if _native.device_count() < 1:
        sys.exit('bench.py needs a GPU (no CPU fallback for the hot path)')
    ctx = _native.Context(int(os.environ.get('LOCAL_RANK', '0')))
    k, L = args.k, args.read_len
    n_reads = args.reads
    nbytes = n_reads * (L + 1)
    dev_buf = ctx.alloc(nbytes)
    ctx.synth_reads_device(2, 0, n_reads, L, dev_buf)     # seed 2
    ctx.sync()
    r = count_measure(ctx, k, dev_buf, nbytes, n_reads, L, args.strategy, args.steps, args.warmup)
    line = {
        'metric': 'Gbases/s k-mer counted (k=%d, %dbp synthetic)' % (k, L), 'value': r['value'], 'unit': 'Gbases/s',
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': r['ms_per_step'],
        'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None, 'dtype': 'int64', 'data': 'synthetic',
        'config': {'workload': 'k=%d, %d synthetic %dbp reads per GPU resident in HBM, count+balance' % (k, n_reads, L),
                   'k': k, 'reads_per_gpu': n_reads, 'read_len': L, 'strategy': args.strategy, 'pipeline': r['plan'],
                   'parallelism': 'one GPU'},
        'checksum_ok': r['checksum_ok'], 'rccl_ranks': 1, 'src_sha': source_sha(), 'roofline': r['roofline'],
    }
    ok = r['checksum_ok']
    if not args.no_cpu:
        # the private-table plan of the all-cores figure gets a sample that amortises its fixed cost: the first 40 M reads of the
        # device buffer (what the host's memory allows: the sample + 64..128 tables of 4^k entries)
        big = None
        big_reads = min(n_reads, args.cpu_big_reads)
        if big_reads > args.cpu_reads and k <= 12:
            big = np.empty(big_reads * (L + 1), dtype=np.uint8)
            ctx.d2h(big, dev_buf)
        line['cpu_baseline'] = cpu_baseline(k, L, args.cpu_reads, big=big)
        del big
    if not args.no_extra and k == 12 and args.strategy == 'auto':
        line['extra'] = ex = run_extras(ctx, args, dev_buf, nbytes, r['ms_per_step'])
        # the other BASELINE configs and the host-resident figures as SCALARS of `config` (the driver's record keeps scalars only)
        for key, src, field in (('config4_k15_ms_per_step', 'k15', 'ms_per_step'), ('config4_k15_Gbases_per_s', 'k15', 'value'),
                                ('config5_matrix_prod_ms', 'matrix_prod', 'ms_per_step'), ('config5_matrix_sum_ms', 'matrix_sum', 'ms_per_step'),
                                ('config5_matrix_euclidean_ms', 'matrix_euclidean', 'ms_per_step'),
                                ('host_reads_overlapped_Gbases_per_s', 'end_to_end', 'overlapped_Gbases_per_s'),
                                ('fasta_file_Gbases_per_s', 'fasta_end_to_end', 'Gbases_per_s')):
            if isinstance(ex.get(src), dict) and field in ex[src]:
                line['config'][key] = ex[src][field]
        if isinstance(ex.get('skew'), dict):
            for name, v in ex['skew'].items():
                if isinstance(v, dict) and 'Gbases_per_s' in v:
                    line['config']['skew_%s_Gbases_per_s' % name] = v['Gbases_per_s']
        line['config']['extras_ok'] = all(v.get('checksum_ok', True) is True and 'error' not in v for v in ex.values() if isinstance(v, dict))
    line['config']['checksum_ok'] = r['checksum_ok']
    print(json.dumps(line), flush=True)
    ctx.free(dev_buf)
    ctx.close()
    if not ok:
        sys.exit('checksum mismatch')


if __name__ == '__main__':
    main()
