"""Skewed inputs at FULL size against the oracle (round-4 review, weak 2): the paths such inputs stress are size-dependent --
the spill list and the per-workgroup hot-item table of the quad scatters, FRESH lists from 64 MiB up, pool halving -- and
tools/skewbench.py times 1 GiB of exactly these inputs without comparing them with anything.  Here: the five inputs of
tools/skewbench.py plus reads that share an adapter prefix, reads with poly-A stretches inside and period-3 repeats, 1 GiB each, generated on the device (torch, seeded), counted through
AUTO as a whole device feed (what bench.py times) and compared with the CPU oracle run on the SAME bytes downloaded:
  * k = 12: every one of the 4^12 bins, plain and balanced (oracle.count_flat on private per-thread tables, oracle.balance);
  * k = 15: 64 blocks of 2^20 table entries, plain and balanced (oracle.count_blocks) -- the poly-A / poly-T, (AC)n / (GT)n and
    adapter blocks with the blocks of their reverse complements, the first and last block, the rest spread over the table;
and kpal_count_stats shows that the slow paths really ran: items rode in spill lists, hot-item tables were used, spill lists
overflowed, and (k = 15, a context with small bypass lists) a FRESH piece was run again after a list overflow."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

L = 151
NBYTES = 1 << 30
READS = NBYTES // L
ADAPTER = b'AGATCGGAAGAGCACACGTCTGAACTCCAGTCAC'       # 34 bases, a TruSeq-like prefix shared by every read


def make_cases(torch, dev):
    """name -> uint8 cuda tensor of READS * 151 bytes (150 bases + '\\n' per read)."""
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    acgt = torch.tensor(list(b'ACGT'), dtype=torch.uint8, device=dev)

    def sample(p, n):
        """n bases with P(A, C, G, T) = p (p: 4 floats, or a float32 tensor [n, 3] of cumulative thresholds)."""
        u = torch.rand(n, generator=g, device=dev)
        if isinstance(p, (list, tuple)):
            t1, t2, t3 = p[0], p[0] + p[1], p[0] + p[1] + p[2]
            code = (u >= t1).to(torch.uint8) + (u >= t2).to(torch.uint8) + (u >= t3).to(torch.uint8)
        else:
            code = (u >= p[:, 0]).to(torch.uint8) + (u >= p[:, 1]).to(torch.uint8) + (u >= p[:, 2]).to(torch.uint8)
        return acgt[code.long()]

    def as_reads(flat):
        m = flat[:READS * L].view(READS, L)
        m[:, L - 1] = 10
        return m

    cases = {}
    cases['uniform'] = as_reads(sample([.25, .25, .25, .25], READS * L)).reshape(-1)
    cases['at_rich'] = as_reads(sample([.32, .18, .18, .32], READS * L)).reshape(-1)
    low = as_reads(sample([.25, .25, .25, .25], READS * L)).clone()
    hit = torch.rand(READS, generator=g, device=dev) < 0.02          # 2 % of the reads are poly-A / (AC)n
    poly = torch.rand(READS, generator=g, device=dev) < 0.5
    ac = torch.tensor(list(b'AC' * 75), dtype=torch.uint8, device=dev)
    low[:, :150] = torch.where((hit & poly)[:, None], torch.full_like(low[:, :150], ord('A')), low[:, :150])
    low[:, :150] = torch.where((hit & ~poly)[:, None], ac[None, :].expand(READS, 150), low[:, :150])
    cases['low_complexity_2pct'] = low.reshape(-1)
    # one long record, GC content drifting between 35 % and 55 % over 2 MiB windows
    n = READS * L
    win = torch.arange(n, device=dev) >> 21
    gc = 0.45 + 0.10 * torch.sin(win.float() * 0.7)
    thr = torch.stack([(1 - gc) / 2, (1 - gc) / 2 + gc / 2, (1 - gc) / 2 + gc], dim=1)
    cases['drifting_gc_one_record'] = sample(thr, n)
    del win, gc, thr
    cases['homopolymer'] = torch.full((n,), ord('A'), dtype=torch.uint8, device=dev)
    ad = as_reads(sample([.25, .25, .25, .25], READS * L)).clone()
    ad[:, :len(ADAPTER)] = torch.tensor(list(ADAPTER), dtype=torch.uint8, device=dev)[None, :]
    cases['adapter_prefixed'] = ad.reshape(-1)
    # poly-A stretches INSIDE reads, flanked by valid bases (RNA-seq tails): the two flank items per stretch land in the poly-A row with
    # DISTINCT values -- singletons in their wave whose k-mers (A^11 X ...) repeat over the whole input: the k-mer entries of the
    # hot-item table and their ageing; and period-3 repeats, which are no repeat lanes (three items cycle)
    inside = as_reads(sample([.25, .25, .25, .25], READS * L)).clone()
    hit2 = torch.rand(READS, generator=g, device=dev) < 0.02
    inside[:, 5:133] = torch.where(hit2[:, None], torch.full_like(inside[:, 5:133], ord('A')), inside[:, 5:133])
    cases['polya_inside_2pct'] = inside.reshape(-1)
    p3 = as_reads(sample([.25, .25, .25, .25], READS * L)).clone()
    acg = torch.tensor(list(b'ACG' * 50), dtype=torch.uint8, device=dev)
    p3[:, :150] = torch.where(hit2[:, None], acg[None, :].expand(READS, 150), p3[:, :150])
    cases['period3_2pct'] = p3.reshape(-1)
    return cases


def prefix_block(seq5):
    return int(oracle.from_sequences([seq5], 5).argmax())


@pytest.fixture(scope='module')
def cases():
    torch = pytest.importorskip('torch')
    return make_cases(torch, 'cuda:0')


def test_full_size_skewed_k12(cases):
    from kpal_amd import _native
    import torch
    k = 12
    ctx = _native.Context(0)
    threads = min(32, os.cpu_count() or 1)
    try:
        seen = {}
        for name, t in cases.items():
            before = ctx.count_stats()
            host = t.cpu().numpy()
            torch.cuda.synchronize()
            ctx.count_begin(k)
            ctx.count_feed_device(t.data_ptr(), t.numel())
            plan = ctx.count_last_plan()
            got = ctx.count_finish()
            ctx.count_begin(k)
            ctx.count_feed_device(t.data_ptr(), t.numel())
            ctx.count_balance()                                     # the fused finalisation + balance of the benchmarked step
            got_bal = ctx.count_finish()
            after = ctx.count_stats()
            want = oracle.count_flat(host, k, threads=threads, mode='private')
            np.testing.assert_array_equal(got, want, err_msg='%s plain, plan %r' % (name, plan))
            np.testing.assert_array_equal(got_bal, oracle.balance(want, k), err_msg='%s balanced, plan %r' % (name, plan))
            seen[name] = (plan, {key: after[key] - before[key] for key in after})
            del host, want, got, got_bal
        print('k=12 plans and slow-path statistics (two counts per case):', seen)
        assert seen['uniform'][0][0] == 'partition_quads' and seen['uniform'][0][1] == 8      # the benchmarked tile
        assert seen['uniform'][1]['spilled_items'] > 0                                          # Poisson tails ride in the spill list
        assert seen['uniform'][1]['repeat_pieces'] == 0 and seen['homopolymer'][1]['repeat_pieces'] == 2   # the scatter instantiation follows the sample
        assert seen['low_complexity_2pct'][1]['repeat_pieces'] == 2
        for name in ('homopolymer', 'low_complexity_2pct'):
            assert seen[name][0][0] == 'partition_quads', seen[name]
            assert seen[name][1]['hot_entries'] > 0, seen[name]                                # over-full rows: counted in the hot-item tables
        # the repeat lanes of the homopolymer never reach the rows: nothing spills (round 4: 33 M items in the lists, 503 M beyond them)
        assert seen['homopolymer'][1]['spilled_items'] == 0 and seen['homopolymer'][1]['unlisted_items'] == 0
        # ... and the chain they took before -- row, spill list, carried, hot-item table / beyond the list -- still counts right:
        # the instantiation without the shortcut (KPAL_QUAD_REPEAT=0) on the same 1 GiB
        os.environ['KPAL_QUAD_REPEAT'] = '0'
        try:
            plain = _native.Context(0)
        finally:
            os.environ.pop('KPAL_QUAD_REPEAT', None)
        try:
            for name in ('homopolymer', 'low_complexity_2pct'):
                t = cases[name]
                before = plain.count_stats()
                plain.count_begin(k)
                plain.count_feed_device(t.data_ptr(), t.numel())
                got = plain.count_finish()
                after = plain.count_stats()
                np.testing.assert_array_equal(got, oracle.count_flat(t.cpu().numpy(), k, threads=threads, mode='private'), err_msg=name + ' without the repeat shortcut')
                assert after['repeat_pieces'] == before['repeat_pieces'] and after['spilled_items'] > before['spilled_items'], name
                if name == 'homopolymer':
                    assert after['unlisted_items'] > before['unlisted_items']      # more items than the spill list holds
        finally:
            plain.close()
        assert seen['adapter_prefixed'][0][0] in ('partition_chunked', 'partition_quads')       # (AUTO: a spread hot excess goes to the chunked pipeline)
    finally:
        ctx.close()


# k = 15: the first quarter of every case (256 MiB of whole reads).  The paths at stake -- FRESH lists (first whole-buffer piece of
# at least max(64 MiB, 4^k / 8 B = 128 MiB)), their overflow and the classic rerun, level-1 / level-2 spill lists and hot-item
# tables -- are the same from that size on, and eight cases x (three 8 GiB tables + the host oracle over the downloaded bytes) at
# 1 GiB took three minutes of the GPU suite's time box (round 5: 141 s, round 6: 181 s).  k = 12 stays at 1 GiB.
NBYTES_K15 = (NBYTES // 4) // L * L


def test_full_size_skewed_k15(cases):
    from kpal_amd import _native, dist
    import torch
    k, block_bits = 15, 20
    sel = [0, 1023, prefix_block('ACACA'), prefix_block('CACAC'), prefix_block('TGTGT'), prefix_block('GTGTG'),
           prefix_block('AGATC'), prefix_block('GATCT'), prefix_block('GTCAC'), prefix_block('GTGAC'), 341, 682, 1, 1022, 512, 511]
    rs = np.random.RandomState(15)
    sel = list(dict.fromkeys(sel))
    sel += [int(b) for b in rs.permutation(1024) if int(b) not in sel][:64 - len(sel)]
    assert len(set(sel)) == 64
    ctx = _native.Context(0)
    os.environ['KPAL_DIRECT_SEG'] = '64'
    try:
        small = _native.Context(0)                                 # bypass lists of 64 entries: a skewed FRESH piece overflows them
    finally:
        os.environ.pop('KPAL_DIRECT_SEG', None)
    sel_t = torch.as_tensor(sel, device='cuda:0')
    try:
        seen = {}
        for name, whole in cases.items():
            t = whole[:NBYTES_K15]
            before = ctx.count_stats()
            host = t.cpu().numpy()
            torch.cuda.synchronize()
            ctx.count_begin(k)
            ctx.count_feed_device(t.data_ptr(), t.numel())
            plan = ctx.count_last_plan()
            ctx.count_finish(to_host=False)
            ctx.sync()
            full = dist.table_as_tensor(ctx)
            total = int(full.sum())
            blocks = full.view(-1, 1 << block_bits)[sel_t].cpu().numpy()
            torch.cuda.synchronize()
            ctx.count_begin(k)
            ctx.count_feed_device(t.data_ptr(), t.numel())
            ctx.count_balance()
            ctx.count_finish(to_host=False)
            ctx.sync()
            bal = dist.table_as_tensor(ctx)
            assert int(bal.sum()) == 2 * total
            bal_blocks = bal.view(-1, 1 << block_bits)[sel_t].cpu().numpy()
            torch.cuda.synchronize()
            after = ctx.count_stats()
            # (one hot bin: a shared table with atomic adds serialises the host threads -- few threads there)
            threads = 2 if name == 'homopolymer' else min(32, os.cpu_count() or 1)
            plain, mirror = oracle.count_blocks(host, k, block_bits, sel, threads=threads)
            assert np.array_equal(blocks, plain), '%s plain, plan %r' % (name, plan)
            mirror += plain
            assert np.array_equal(bal_blocks, mirror), '%s balanced, plan %r' % (name, plan)
            seen[name] = (plan, total, {key: after[key] - before[key] for key in after})
            if name in ('low_complexity_2pct', 'adapter_prefixed'):
                # the same feed on the context with tiny bypass lists: FRESH, overflow, run again the classic way -- same table
                b2 = small.count_stats()
                small.count_begin(k, 'partition2_quads')
                small.count_feed_device(t.data_ptr(), t.numel())
                small.count_balance()
                small.count_finish(to_host=False)
                small.sync()
                a2 = small.count_stats()
                assert torch.equal(dist.table_as_tensor(small), bal), name
                torch.cuda.synchronize()
                seen[name + '/small_lists'] = {key: a2[key] - b2[key] for key in a2}
            del host, plain, mirror, blocks, bal_blocks
        print('k=15 plans, totals and slow-path statistics (two counts per case):', seen)
        assert seen['uniform'][0][0] == 'partition2_quads' and seen['uniform'][2]['fresh_pieces'] == 2
        assert seen['uniform'][2]['spilled_items'] > 0
        assert seen['homopolymer'][1] == NBYTES_K15 - k + 1
        assert seen['homopolymer'][2]['hot_entries'] > 0, seen['homopolymer']
        assert any(seen[n + '/small_lists']['fresh_reruns'] >= 1 for n in ('low_complexity_2pct', 'adapter_prefixed')), seen
    finally:
        small.close()
        ctx.close()
