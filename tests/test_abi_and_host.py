"""CPU-side checks: the C-ABI library loads and exports every symbol include/kpal_hip.h
declares, the product path fails loudly without a GPU, host logic (FASTA tokenising, names,
sharding) and the world_size-2 gloo reduce."""
import io
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__
    __graft_entry__.build()
    from kpal_amd import _native
    return _native


def test_header_symbols_exported(built):
    header = open(os.path.join(ROOT, 'include', 'kpal_hip.h')).read()
    declared = set(re.findall(r'\b(kpal_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 25
    L = built.load()
    for name in sorted(declared):
        assert hasattr(L, name), 'libkpal_hip.so does not export %s' % name
    assert declared == set(built.SIGNATURES), declared ^ set(built.SIGNATURES)
    assert b'gfx950' in L.kpal_version()


def test_host_helpers_without_gpu(built):
    import oracle
    for k in (1, 2, 5, 12, 15, 16):
        for x in (0, 1, 4 ** k - 1, 12345 % 4 ** k):
            assert built.reverse_complement(x, k) == oracle.reverse_complement(x, k)


def test_fails_loudly_without_gpu(built):
    if built.device_count() > 0:
        pytest.skip('a GPU is visible')
    from kpal_amd import klib, metrics
    with pytest.raises(RuntimeError):
        klib.Profile.from_sequences(['ACGT'], 2)
    with pytest.raises(RuntimeError):
        metrics.multiset(np.ones(4, dtype=np.int64), np.ones(4, dtype=np.int64), metrics.pairwise['prod'])
    with pytest.raises(RuntimeError):
        klib.Profile(np.ones(16, dtype=np.int64)).balance()


def test_no_oracle_import_in_product():
    for top in ('kpal_amd', 'kpal'):   # (include/kpal_hip.h only NAMES the oracle's generator in a comment)
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith(('.py', '.hip', '.hpp', '.h')):
                    text = open(os.path.join(dirpath, f)).read()
                    assert 'import oracle' not in text and 'from oracle' not in text and 'kpal_oracle' not in text, f


def test_import_kpal_is_the_drop_in():
    """`import kpal` (the reference's package name) resolves to kpal_amd's modules: stock callers run unmodified with this
    repository first on sys.path -- `from kpal import klib`, `from kpal.kdistlib import ProfileDistance`, `kpal.kmer.main`."""
    code = ('import sys; sys.path.insert(0, %r)\n'
            'import kpal, kpal.klib, kpal.kmer, kpal_amd\n'
            'from kpal.kdistlib import ProfileDistance, distance_matrix\n'
            'from kpal import metrics, ProfileFileType\n'
            'assert kpal.klib is kpal_amd.klib and kpal.kmer.main is kpal_amd.kmer.main\n'
            'assert ProfileDistance is kpal_amd.kdistlib.ProfileDistance and metrics.multiset is kpal_amd.metrics.multiset\n'
            'print("SHIM_OK")\n') % ROOT
    p = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert p.returncode == 0 and b'SHIM_OK' in p.stdout, p.stdout.decode()[-2000:]
    p = subprocess.run([sys.executable, '-m', 'kpal', '--help'], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert p.returncode == 0 and b'count' in p.stdout and b'matrix' in p.stdout


def test_fasta_tokeniser_and_profile_surface():
    from kpal_amd import klib
    recs = list(klib._fasta_records(io.StringIO('junk\n>a first\nACG\nT G\n\n>b\n>\nNN\n')))
    assert recs == [('a', 'ACGTG'), ('b', ''), ('', 'NN')]
    p = klib.Profile(np.arange(16, dtype=np.int64), 'n')
    assert (p.length, p.number, p.total, p.non_zero) == (2, 16, 120, 15)
    assert p.binary_to_dna(p.dna_to_binary('GT')) == 'GT'
    with pytest.raises(KeyError):
        p.dna_to_binary('AN')
    with pytest.raises(ValueError):
        p.name = 'a.b'
    p.name = None
    q = p.copy()
    q.counts[0] = 99
    assert p.counts[0] == 0
    with pytest.raises(ValueError):
        klib.Profile(np.arange(4, dtype=np.int64)).shrink(1)       # factor must be smaller than k
    p.merge(q, merger=lambda x, y: x + 2 * y)                       # user callable: NumPy, as the reference
    assert p.counts[0] == 198 and p.counts[1] == 3


def test_metrics_numpy_helpers():
    from kpal_amd import metrics
    assert set(metrics.pairwise) == {'prod', 'sum'} and set(metrics.mergers) == {'sum', 'xor', 'int', 'nint'}
    assert set(metrics.summary) == {'min', 'average', 'median'} and set(metrics.vector_distance) == {'default', 'euclidean', 'cosine'}
    l, r = np.array([1, 2, 3]), np.array([2, 4, 6])
    assert metrics.get_scale(l, r) == (2.0, 1.0) and metrics.scale_down(2.0, 1.0) == (1.0, 0.5)
    assert list(metrics.positive(l, [0, 1, 1])) == [0, 2, 3]
    assert metrics.distribution([1, 1, 2]) == [(1, 2), (2, 1)]
    np.testing.assert_allclose(metrics.pairwise['prod'](l, r), abs(l - r) / ((l + 1) * (r + 1)))


def test_shard_range():
    from kpal_amd import dist
    for n in (0, 1, 7, 100, 10 ** 8):
        for w in (1, 2, 3, 8):
            blocks = [dist.shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (f0, c0), (f1, _) in zip(blocks[:-1], blocks[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        dist.shard_range(10, 2, 2)


def fasta_corpus(tmp_path):
    """FASTA files that stress the shard cutter: -> {name: path}.  Many short records; one giant record in 60-column lines;
    one giant record on ONE line (cuts fall inside the line); CRLF line ends; text before the first header, blank lines,
    blanks inside and at the end of lines, an empty record, lower case and N runs; the tutorial files."""
    import oracle
    rs = np.random.RandomState(7)
    acgt = np.frombuffer(b'ACGT', dtype=np.uint8)

    def seq(n, noisy=False):
        s = acgt[rs.randint(0, 4, n)].copy()
        if noisy and n > 50:
            for _ in range(n // 200 + 1):
                at = rs.randint(0, n - 8)
                s[at:at + rs.randint(1, 8)] = ord('N')
            low = rs.randint(0, n, n // 50)
            s[low] |= 0x20
        return s.tobytes().decode()

    def wrap(s, width=60, eol='\n'):
        return eol.join(s[i:i + width] for i in range(0, len(s), width)) + eol

    files = {}
    files['many'] = ''.join('>r%d some description\n%s' % (i, wrap(seq(rs.randint(0, 700), True))) for i in range(400))
    files['giant_wrapped'] = '>chr1\n' + wrap(seq(300000, True))
    files['giant_one_line'] = '>chrU\n' + seq(250000, True) + '\n>tail\nACGTTGCA\n'
    files['crlf'] = ''.join('>c%d\r\n%s' % (i, wrap(seq(rs.randint(1, 500)), 70, '\r\n')) for i in range(150))
    files['messy'] = ('no header yet\nACGTACGTACGT\n\n>m1  x\nAC GT\tAC \nGGTT\t\n\n  \nACGTAC\n>\n>m3\n' + wrap(seq(5000, True), 33) +
                      '>m4\n' + wrap(seq(20000), 80) + '\n>m5\nAC\n')
    files['no_record'] = 'just text\nACGTACGT\n'
    out = {}
    for name, text in files.items():
        p = tmp_path / (name + '.fa')
        p.write_bytes(text.encode('latin-1'))
        out[name] = str(p)
    tut = os.path.join(ROOT, 'tests', 'golden', 'tutorial')
    for f in sorted(os.listdir(tut))[:3]:
        if f.endswith('.fa'):
            out['tutorial_' + f] = os.path.join(tut, f)
    return out


def records_of(path):
    from kpal_amd import klib
    with open(path, encoding='latin-1', newline='') as fh:
        return [s for _, s in klib._fasta_records(io.StringIO(fh.read().replace('\r\n', '\n').replace('\r', '\n')))]


def test_fasta_shards_tile_the_input_and_count_every_window_once(tmp_path, monkeypatch):
    """kpal_amd.dist.fasta_shards (north_star: FASTA shards across the GPUs): for world sizes 1..9 and several k
    the shards' byte ranges tile the files exactly, every range begins at a header line or carries the halo prefix of a cut
    inside a record, and the SUM of the per-shard counts -- each shard tokenised and counted on its own, the oracle standing
    in for the GPU -- equals the count of all records of all files (kpal/klib.py:97-112 per file, merged by `sum`:
    doc/tutorial.rst:94-95)."""
    import oracle
    from kpal_amd import dist, klib
    corpus = fasta_corpus(tmp_path)
    monkeypatch.setattr(dist, '_LONG_LINE', 1000)      # (64 KiB: a cut moves back to its line's start unless that is further away)
    cases = [[corpus[n]] for n in sorted(corpus)] + [[corpus['many'], corpus['giant_one_line'], corpus['no_record'], corpus['messy']]]
    for paths in cases:
        for k in (1, 2, 5, 12):
            want = oracle.from_sequences([s for p in paths for s in records_of(p)], k)
            for world in (1, 2, 3, 4, 7, 9):
                shards = dist.fasta_shards(paths, world, k)
                assert len(shards) == world
                # the ranges tile every file, in order
                covered = {p: 0 for p in paths}
                for segs in shards:
                    for seg in segs:
                        assert seg.begin == covered[seg.path] and seg.end > seg.begin, (paths, world, seg)
                        covered[seg.path] = seg.end
                assert all(covered[p] == os.path.getsize(p) for p in paths)
                got = np.zeros(4 ** k, dtype=np.int64)
                for segs in shards:
                    for seg in segs:
                        assert len(seg.prefix) <= 2 + max(k - 1, 0)
                        text = dist.segment_text(seg).decode('latin-1').replace('\r\n', '\n').replace('\r', '\n')
                        got += oracle.from_sequences([s for _, s in klib._fasta_records(io.StringIO(text))], k)
                np.testing.assert_array_equal(got, want, err_msg='%r k=%d world=%d' % (paths, k, world))
    # sizes are balanced where the text allows it: one giant record is cut inside
    for name in ('giant_wrapped', 'giant_one_line'):
        shards = dist.fasta_shards(corpus[name], 4, 12)
        sizes = [sum(s.end - s.begin for s in segs) for segs in shards]
        assert max(sizes) < 1.3 * min(sizes), (name, sizes)
        assert any(seg.prefix for segs in shards for seg in segs)


_GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as td
import oracle
from kpal_amd import dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
td.init_process_group('gloo', rank=rank, world_size=world)
k, n_reads = 6, 1001
first, n = dist.shard_range(n_reads, rank, world)
local = oracle.count_flat(oracle.synth_reads(3, first, n, 150), k)      # per-rank table (oracle stands in for the GPU here)
t = torch.from_numpy(local.copy())
dist.reduce_counts(t, dst=0)
if rank == 0:
    want = oracle.count_flat(oracle.synth_reads(3, 0, n_reads, 150), k)
    assert np.array_equal(t.numpy(), want), 'sharded + reduced counts differ from the single-stream count'
    print('GLOO_OK', int(t.sum()))
# the reducer bench.py uses, every flag combination: three steps over different shards of reads; the merged
# + balanced table of every step must equal balance(single-stream count), whichever dtype travelled
for mode in ('int64', 'u32'):
    for overlap in (False, True):
        for big in (False, True):          # big: one bin near 2^31 on rank 1 forces the int64 fall-back of the u32 mode
            table = torch.zeros(4 ** k, dtype=torch.int64)
            merged = []
            red = dist.TableReducer(table, balance=lambda t: t.copy_(torch.from_numpy(oracle.balance(t.numpy().copy(), k))),
                                    mode=mode, overlap=overlap)
            wants = []
            for step in range(3):
                if overlap and step:        # the previous step's result is valid until the next reduce_step starts
                    red.drain()
                    if rank == 0:
                        merged.append(red.result().numpy().copy())
                mine = oracle.count_flat(oracle.synth_reads(30 + step, first, n, 150), k)
                if big and rank == 1:
                    mine[5] += 2 ** 31 - 7
                table.copy_(torch.from_numpy(mine))
                red.reduce_step()
                if not overlap and rank == 0:
                    merged.append(red.result().numpy().copy())
                w = oracle.count_flat(oracle.synth_reads(30 + step, 0, n_reads, 150), k)
                if big:
                    w[5] += 2 ** 31 - 7
                wants.append(oracle.balance(w, k))
            red.drain()
            if overlap and rank == 0:
                merged.append(red.result().numpy().copy())
            if rank == 0:
                assert len(merged) == 3
                for got, want in zip(merged, wants):
                    assert np.array_equal(got, want), (mode, overlap, big)
                if mode == 'u32':
                    assert (red.steps_u32, red.steps_int64) == ((0, 3) if big else (3, 0)), (red.steps_u32, red.steps_int64)
                else:
                    assert red.steps_u32 == 0
if rank == 0:
    print('REDUCER_OK')
# FASTA shards (kpal_amd.dist.fasta_shards): every rank tokenises and counts ITS byte ranges, one reduce, rank 0 compares
# with the count of all records of all files
import io
from kpal_amd import klib
paths = %(fasta)r
def records(text):
    return [s for _, s in klib._fasta_records(io.StringIO(text.decode('latin-1').replace('\\r\\n', '\\n').replace('\\r', '\\n')))]
for k in (4, 9):
    mine = np.zeros(4 ** k, dtype=np.int64)
    for seg in dist.fasta_shards(paths, world, k)[rank]:
        mine += oracle.from_sequences(records(dist.segment_text(seg)), k)
    t = torch.from_numpy(mine)
    dist.reduce_counts(t, dst=0)
    if rank == 0:
        want = oracle.from_sequences([s for p in paths for s in records(open(p, 'rb').read())], k)
        assert np.array_equal(t.numpy(), want), 'FASTA shards: merged counts differ from the whole-file count (k=%%d)' %% k
if rank == 0:
    print('FASTA_SHARDS_OK')
# the bin-range merge (reduce-scatter + mirrored-range exchange, kpal_amd.dist.reduce_scatter_balance): every rank's range of the
# merged + balanced table against oracle.balance of the single-stream count
for k in (4, 7):
    mine = oracle.count_flat(oracle.synth_reads(50 + k, first, n, 150), k)
    t = torch.from_numpy(mine.copy())
    lo, cnt = dist.reduce_scatter_balance(t, k, balance=True)
    want = oracle.balance(oracle.count_flat(oracle.synth_reads(50 + k, 0, n_reads, 150), k), k)
    assert (lo, cnt) == (rank * 4 ** k // world, 4 ** k // world)
    assert np.array_equal(t.numpy()[lo:lo + cnt], want[lo:lo + cnt]), 'bin-range merge: rank %%d range differs (k=%%d)' %% (rank, k)
    t2 = torch.from_numpy(mine.copy())
    dist.reduce_scatter_balance(t2, k, balance=False)
    plain = oracle.count_flat(oracle.synth_reads(50 + k, 0, n_reads, 150), k)
    assert np.array_equal(t2.numpy()[lo:lo + cnt], plain[lo:lo + cnt])
ok = torch.ones(1, dtype=torch.int64)
td.all_reduce(ok)
if rank == 0 and int(ok.item()) == world:
    print('RANGE_MERGE_OK')
td.barrier()
td.destroy_process_group()
'''


_GLOO_RANGE_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as td
import oracle
from kpal_amd import dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
td.init_process_group('gloo', rank=rank, world_size=world)
n_reads = 803
first, n = dist.shard_range(n_reads, rank, world)
for k in (3, 6, 8):
    mine = oracle.count_flat(oracle.synth_reads(70 + k, first, n, 150, noisy=True), k)
    t = torch.from_numpy(mine.copy())
    lo, cnt = dist.reduce_scatter_balance(t, k, balance=True)
    want = oracle.balance(oracle.count_flat(oracle.synth_reads(70 + k, 0, n_reads, 150, noisy=True), k), k)
    assert np.array_equal(t.numpy()[lo:lo + cnt], want[lo:lo + cnt]), (rank, k)
ok = torch.ones(1, dtype=torch.int64)
td.all_reduce(ok)
if rank == 0 and int(ok.item()) == world:
    print('RANGE_MERGE_OK')
td.barrier()
td.destroy_process_group()
'''


def test_world_size_4_gloo_bin_range_merge(tmp_path):
    """The bin-range merge at world 4 (an even number of range bits; world 2 -- an odd one -- runs in
    test_world_size_2_gloo_reduce): reduce_scatter + one all_to_all of the mirrored entries, every rank's range against the
    oracle's balance of the single-stream count (the oracle stands in for the per-rank GPU count)."""
    script = tmp_path / 'range_worker.py'
    script.write_text(_GLOO_RANGE_WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29547', WORLD_SIZE='4')
    procs = []
    for rank in range(4):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'RANGE_MERGE_OK' in outs[0], outs[0]


def test_range_index_arithmetic(tmp_path):
    """CPU emulation of an index scheme the device kernels share their header with: the bin-range merge of per-rank tables
    (csrc/range_index.hpp: W ranks, reduce-scatter, pack, all-to-all, unpack against balance of the summed table; k = 2..8 for
    every W, packing bijection at k = 13..16)."""
    import shutil
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    for name, token in (('range_index_check', 'RANGE_INDEX_OK'),):
        exe = str(tmp_path / name)
        b = subprocess.run(['g++', '-O2', '-std=c++17', '-o', exe, os.path.join(ROOT, 'tests', 'native', name + '.cpp')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert b.returncode == 0, b.stdout.decode()[-3000:]
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert r.returncode == 0 and token in r.stdout.decode(), r.stdout.decode()[-3000:]


def test_world_size_2_gloo_reduce(tmp_path):
    """N > 1 path on CPU: shard reads over 2 ranks, reduce the int64 tables with one collective
    (gloo here, RCCL on the GPUs), compare with the single-stream count; the bin-range merge at world 2."""
    script = tmp_path / 'worker.py'
    corpus = fasta_corpus(tmp_path)
    fasta = [corpus['many'], corpus['giant_wrapped'], corpus['messy'], corpus['giant_one_line']]
    script.write_text(_GLOO_WORKER % {'root': ROOT, 'fasta': fasta})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', WORLD_SIZE='2')
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'GLOO_OK %d' % (1001 * 145) in outs[0]
    assert 'REDUCER_OK' in outs[0], outs[0]
    assert 'FASTA_SHARDS_OK' in outs[0], outs[0]
    assert 'RANGE_MERGE_OK' in outs[0], outs[0]


def test_quad2_index_arithmetic(tmp_path):
    """The index arithmetic of the two-level quad pipeline's finalisation (kpal_amd/csrc/quad2_index.hpp: staging layout,
    reverse-complement-closed sets, stream orders, partner positions) on the host: the functions the device kernels use
    drive a CPU emulation of staging + finalisation (plain and balancing) at k = 13 over random forms and tables, compared
    with v = T + forms and out[i] = v[i] + v[rc(i)]; single sets incl. self-paired ones for k = 13..16
    (tests/native/quad2_index_check.cpp)."""
    import shutil
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    exe = str(tmp_path / 'quad2_index_check')
    b = subprocess.run(['g++', '-O2', '-std=c++17', '-o', exe, os.path.join(ROOT, 'tests', 'native', 'quad2_index_check.cpp')],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert b.returncode == 0, b.stdout.decode()[-3000:]
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0 and 'QUAD2_INDEX_OK' in r.stdout.decode(), r.stdout.decode()[-3000:]


def test_bench_launcher_spawns_the_ranks():
    """`python bench.py --gpus 2` as a PLAIN command (how the driver starts the scaling runs): the launcher must start the
    two ranks itself through torch.distributed.run, before anything touches a GPU, relay rank 0's JSON line and exit with
    the ranks' status.  --stub replaces the GPU work by CPU tensors over gloo (plumbing only, value null)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--stub'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['checksum_ok'] is True and line['steps'] == 3
    assert line['merged_equals_single_stream'] is True and line['attempt'] == 1
    # a failing rank fails the launcher (both attempts of the supervisors: there is no GPU here)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=dict(env, HIP_VISIBLE_DEVICES='', ROCR_VISIBLE_DEVICES=''), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0


def test_bench_supervisor_keeps_a_measured_headline():
    """A hang AFTER the headline reduce mode has been measured and verified (rank 0 publishes its line at that point) must not cost
    the measurement: the supervisors end the attempt, rank 0's prints the published line with `later_modes_failed`, nobody starts
    a second attempt, the job exits 0."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(KPAL_BENCH_RUN_TIMEOUT='4', KPAL_BENCH_STARTUP_TIMEOUT='120')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '0', '--stub', '--stub-hang-after-headline'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line['attempt'] == 1 and line['merged_equals_single_stream'] is True and 'did not finish within' in line['later_modes_failed']
    assert 'attempt 2' not in p.stderr.decode()


@pytest.mark.parametrize('fault', ['hang', 'rank_dies'])
def test_bench_supervisor_falls_back_to_a_fresh_worker(fault):
    """A first multi-GPU attempt that never returns must still end in a line: every rank under torch.distributed.run is a
    supervisor that starts the GPU worker as a child, and on a worker that does not finish in time (hang: every worker of
    attempt 1 stops responding after start-up) or dies (rank_dies: rank 1's worker exits before the rendezvous while rank 0's
    waits for it) ends the attempt on ALL ranks -- failure flags in the job's shared directory -- kills the workers' process
    groups and starts FRESH workers in the conservative mode (--reduce-via torch --serial-reduce, a new rendezvous).  Driven
    with the CPU/gloo stub."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(KPAL_BENCH_RUN_TIMEOUT='4', KPAL_BENCH_STARTUP_TIMEOUT='120')
    extra = ['--stub-hang-attempt', '1'] if fault == 'hang' else ['--stub-fail-rank', '1']
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '0', '--stub'] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = p.stderr.decode()
    assert p.returncode == 0, err[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line['attempt'] == 2 and line['n_gpus'] == 2 and line['merged_equals_single_stream'] is True
    if fault == 'hang':
        assert 'did not finish within' in line['fallback_reason'], line
    else:
        assert 'exited with status 3' in line['fallback_reason'] or 'gave up on attempt 1' in line['fallback_reason'], line
        assert time.time() - t0 < 120          # the peer's flag ends rank 0's wait at once, not its time limit
    assert 'supervisor (rank' in err


def test_kmer_caller_host_logic():
    """kpal_amd.kmer (kpal/kmer.py:41-48,112-146,541-700): handle names, custom functions, and the argument
    errors that are raised before any profile is touched."""
    import memh5
    from kpal_amd import kmer

    class Named(io.StringIO):
        name = '/data/run7/sample_A.fasta'

    class Stdin(io.StringIO):
        name = '<stdin>'

    assert kmer._name_from_handle(Named()) == 'sample_A'
    assert kmer._name_from_handle(Stdin()) is None and kmer._name_from_handle(io.StringIO()) is None
    f = kmer._custom_function('np.maximum(left, right) + 1', 'left, right')
    assert list(f(np.array([1, 5]), np.array([3, 2]))) == [4, 6]
    assert kmer._custom_function('numpy.minimum', 'left, right') is np.minimum
    with pytest.raises(ValueError, match='number of profile names does not match'):
        kmer.count([io.StringIO('>a\nAC\n')], memh5.File(), 2, names=['a', 'b'])
    empty = memh5.File()
    with pytest.raises(ValueError, match='at least two'):
        kmer.distance_matrix(empty, io.StringIO())
    one = memh5.File()
    one.create_dataset('profiles/x', data=np.zeros(16), dtype='int64', compression='gzip')
    with pytest.raises(ValueError, match='left and right profile names do not match'):
        kmer.distance(one, empty, io.StringIO())
    with pytest.raises(ValueError, match='left and right profile names do not match'):
        kmer.merge(one, empty, memh5.File())
    with pytest.raises(KeyError):
        kmer.balance(one, memh5.File(), names=['missing'])


def test_pmc_summary_tool(tmp_path):
    """tools/pmc_summary.py: per-kernel bytes per dispatch from two rocprofv3 counter passes (FETCH_SIZE x2 on
    gfx950 + WRITE_SIZE, both in KB) and the input bytes per launch from the dispatch count of the named kernel."""
    import json
    header = 'Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n'

    def rows(counter, values):
        out = []
        for i, (name, v) in enumerate(values):
            out.append('%d,%d,1,1,1,1,1,1,"%s",256,0,0,8,0,16,%s,%s,0,1\n' % (i, i, name, counter, v))
        return header + ''.join(out)

    scatter = 'void kpal::chunk_scatter_kernel<12>(kpal::Span, unsigned long)'
    hist = 'void kpal::chunk_hist_kernel<15>(kpal::ChunkPool)'
    (tmp_path / 'f').mkdir()
    (tmp_path / 'w').mkdir()
    (tmp_path / 'f' / 'x_counter_collection.csv').write_text(rows('FETCH_SIZE', [(scatter, 1000), (hist, 500), (scatter, 3000)]))
    (tmp_path / 'w' / 'x_counter_collection.csv').write_text(rows('WRITE_SIZE', [(scatter, 100), (hist, 10), (scatter, 300)]))
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), str(tmp_path / 'f'), str(tmp_path / 'w'),
                        '4000000', 'note', 'chunk_scatter'], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr.decode()
    d = json.loads(p.stdout.decode())
    assert d['launches'] == 2 and d['input_bytes_per_launch_avg'] == 2000000.0
    s = d['kernels']['kpal::chunk_scatter_kernel<12>']
    assert s['dispatches'] == 2 and s['FETCH_SIZE'] == 2000.0 and s['WRITE_SIZE'] == 200.0
    assert s['hbm_bytes_per_dispatch_corrected'] == (2 * 2000.0 + 200.0) * 1024
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), str(tmp_path / 'f'), str(tmp_path / 'w'),
                          '4000000', 'note', 'no_such_kernel'], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert bad.returncode != 0


def test_balance_split_of_scaled_profiles_without_gpu():
    """Float counts (a profile multiplied by a scale factor) never enter the integer kernels:
    balance / split follow the reference's loops (kpal/klib.py:285-327) in NumPy, in the counts'
    dtype, like merge / shrink / the summaries do."""
    import oracle
    from kpal_amd import klib
    rs = np.random.RandomState(3)
    for k in (1, 2, 4):
        c = rs.poisson(5, 4 ** k) * 0.37
        want = c.copy()
        fwd, rev = [], []
        for i in range(4 ** k):
            r = oracle.reverse_complement(i, k)
            if i < r:
                want[i], want[r] = c[i] + c[r], c[r] + c[i]
                fwd.append(c[i] * 2)
                rev.append(c[r] * 2)
            elif i == r:
                want[i] = c[i] + c[i]
                fwd.append(c[i])
                rev.append(c[i])
        p = klib.Profile(c.copy())
        p.balance()
        np.testing.assert_array_equal(p.counts, want)
        f, r_ = klib.Profile(c.copy()).split()
        np.testing.assert_array_equal(f, np.array(fwd))
        np.testing.assert_array_equal(r_, np.array(rev))


def test_fasta_records_follow_seqio_rules():
    """Bio.SeqIO tokenising (kpal/klib.py:111): rstrip per line, ' ' and '\\r' removed, interior tabs stay."""
    from kpal_amd import klib
    text = 'junk\n>r1 desc\nAC\tG T\t\n A\n>\n\tNN \n>r3\n'
    assert list(klib._fasta_records(io.StringIO(text))) == [('r1', 'AC\tGTA'), ('', '\tNN'), ('r3', '')]


def test_command_line_surface_without_gpu(tmp_path, monkeypatch):
    """kpal_amd.kmer.main (kpal/kmer.py:703-975): the seventeen sub-commands with the reference's options and
    defaults, and the usage errors that are raised before any profile is touched (golden G12's wording)."""
    import contextlib
    import json
    import memh5
    from kpal_amd import files, kmer
    parser = kmer.build_parser()
    ns = parser.parse_args(['distance', os.devnull, os.devnull]) if False else None
    # defaults of the reference's front end
    with open(os.path.join(ROOT, 'tests', 'golden', 'cli.json')) as fh:
        g = json.load(fh)['G12']
    store = memh5.Store()
    monkeypatch.setattr(files, 'open_profile_file', store.open)
    monkeypatch.chdir(tmp_path)
    (tmp_path / 'a_1.fa').write_text('>r\nACGT\n')
    (tmp_path / 'a_2.fa').write_text('>r\nACGT\n')
    for name in ('counted.k8', 'merged.k8'):
        h = files.ProfileFileType('w')(name)
        assert h.attrs == {'format': 'kMer', 'version': '1.0.0', 'producer': files.PRODUCER}
    args = parser.parse_args(['count', 'a_1.fa', 'out.k9'])
    assert (args.size, args.by_record, args.names) == (9, False, None)
    args = parser.parse_args(['matrix', 'counted.k8', 'm.txt'])
    assert (args.precision, args.pairwise, args.distance_function, args.summary, args.threshold) == (10, 'prod', 'default', 'min', 0)
    assert not (args.do_balance or args.do_positive or args.do_scale or args.do_smooth or args.down)
    args = parser.parse_args(['shrink', 'counted.k8', 's.k7'])
    assert args.factor == 1
    args = parser.parse_args(['merge', 'counted.k8', 'merged.k8', 'mm.k8'])
    assert (args.merger, args.custom_merger) == ('sum', None)
    wanted = {' '.join(s['argv']): s for s in g['steps'] if s['status']}
    for argv in (['count', '-k', '8', 'a_1.fa', 'counted.k8'], ['nosuchcommand'], ['info', 'a_1.fa'], ['info', 'nosuch.k8'],
                 ['count', '-k', '4', 'a_1.fa', 'a_2.fa', 'bad_names.k4', '-p', 'only_one'],
                 ['cat', 'counted.k8', 'merged.k8', 'bad_prefix.k8', '-x', 'p_']):
        se = io.StringIO()
        with contextlib.redirect_stderr(se), pytest.raises(SystemExit) as exc:
            kmer.main(argv)
        assert exc.value.code == 2
        err = se.getvalue().strip().split('\n')[-1].split('error: ', 1)[1]
        want = wanted[' '.join(argv)]['error']
        if 'Unable to open file' in want or 'invalid choice' in want:
            assert err.split(':')[:2] == want.split(':')[:2]
        else:
            assert err == want
    # versions the reference accepts: >=1.0.0,<2.0.0 (kpal/__init__.py:41)
    assert files.format_version_accepted('1.0.0') and files.format_version_accepted(b'1.4.2') and not files.format_version_accepted('2.0.0')
    h = store.open('counted.k8', 'r')
    h.attrs['version'] = '2.1.0'
    with pytest.raises(Exception, match='not supported'):
        files.ProfileFileType('r')('counted.k8')
    h.attrs['format'] = 'other'
    with pytest.raises(Exception, match='not a k-mer profile file'):
        files.ProfileFileType('r')('counted.k8')
    with pytest.raises(Exception, match='file exists'):
        files.FileType('w')('a_1.fa')


def test_bench_traffic_comes_from_the_committed_profiles():
    """bench.py reports roofline.traffic from profiles/<round>/pmc_hbm_traffic*.json only while those were taken on the kernel sources
    it runs (src_sha); bytes that follow the input are scaled by the input, bytes that follow the 4^k table are not, and every
    kernel of the step -- the balancing finalisation included -- is found in the profile."""
    import bench
    nbytes = 100_000_000 * 151
    here = bench.source_sha()
    for k, names in ((12, ('quad_sample', 'quad_scatter', 'quad_hist', 'quad2_finalize_balanced')),
                     (15, ('quad_sample', 'quad_scatter', 'quad2_scatter', 'quad_hist', 'quad2_finalize_balanced', 'quad2_apply_list'))):
        kernels = {n: (1.0, 1) for n in names}
        info = bench.pmc_traffic(kernels, 'quad_scatter', k, nbytes)
        if info['traffic_profile_src_sha'] != here:
            assert info['traffic'] is None and info['traffic_step'] is None      # a stale profile is never quoted
            continue
        per = info['traffic_by_kernel_per_step']
        assert set(per) == set(names), (k, sorted(per))
        assert 1.9 * nbytes < info['traffic'] < 2.2 * nbytes                     # the scatter reads the input once and writes as much
        table = 8 * 4 ** k
        if k == 15:
            assert 1.4 * table < per['quad2_finalize_balanced'] < 1.6 * table    # forms read (4 B per entry), table written: not scaled
        else:
            assert 2.3 * table < per['quad2_finalize_balanced'] < 2.7 * table    # k = 12 (not FRESH): table read + forms read + table written
            assert per['quad_hist'] < 1.1 * nbytes                               # the records read once, 64 MiB of staged forms written
        assert info['traffic_step'] == pytest.approx(sum(per.values()))
        assert 2.5 * nbytes < info['traffic_step'] < 7 * nbytes


def test_fake_rccl_stand_in_covers_what_the_library_binds(tmp_path):
    """tests/native/fake_rccl.cpp (the transport of the multi-process GPU tests of the kpal_comm_* protocol) builds here, and
    exports every RCCL entry point kpal_multi.hip looks up -- a symbol the library starts to bind must get a stand-in too."""
    src = open(os.path.join(ROOT, 'kpal_amd', 'csrc', 'kpal_multi.hip')).read()
    bound = set(re.findall(r'"(nccl[A-Za-z]+)"', src))
    assert {'ncclCommInitRank', 'ncclReduce', 'ncclReduceScatter', 'ncclSend', 'ncclRecv', 'ncclGroupEnd'} <= bound
    lib = str(tmp_path / 'libfake_rccl.so')
    subprocess.run([os.environ.get('HIPCC', 'hipcc'), '-O2', '-shared', '-fPIC', '-Wall', '-Werror', '-o', lib,
                    os.path.join(ROOT, 'tests', 'native', 'fake_rccl.cpp')], check=True, timeout=600)
    out = subprocess.run(['nm', '-D', '--defined-only', lib], check=True, stdout=subprocess.PIPE).stdout.decode()
    exported = set(re.findall(r' T (nccl[A-Za-z]+)', out))
    assert bound <= exported, bound - exported


def test_the_gatherer_against_its_contract(built):
    """kpal_amd._kpal_gather (csrc/kpal_gather.c: walk and copies on several threads) against a serial restatement of its contract in
    Python -- gather(seq, first, address, cap, threads) copies the items from `first` on, each followed by a newline, while they are
    bytes / bytearray / str of one-byte characters AND fit; -> (next, nbytes, status): 0 the sequence is done, 1 the buffer is full
    (or item `next` alone is longer than it), 2 item `next` is not one it reads -- on random lists and tuples of str / bytes /
    bytearray with latin-1 text, text beyond latin-1, memoryviews, empty items, long items, at capacities from one byte to
    everything and 1..16 threads: the same triple and the same bytes, through the caller's loop to the end of every list."""
    import random
    from kpal_amd import _kpal_gather
    rnd = random.Random(5)

    def restated(seq, first, out, cap):
        at, i = 0, first
        while i < len(seq):
            it = seq[i]
            if isinstance(it, str):
                try:
                    raw = it.encode('latin-1')
                except UnicodeEncodeError:
                    return i, at, 2
            elif type(it) in (bytes, bytearray):
                raw = bytes(it)
            else:
                return i, at, 2
            if at + len(raw) + 1 > cap:
                return i, at, 1
            out[at:at + len(raw)] = np.frombuffer(raw, dtype=np.uint8)
            out[at + len(raw)] = 10
            at += len(raw) + 1
            i += 1
        return i, at, 0

    def item():
        r, n = rnd.random(), rnd.choice([0, 1, 2, 5, 31, 150, 150, 150, 151, 400, 5000])
        body = ''.join(rnd.choice('ACGTNacgt') for _ in range(n))
        if r < 0.4:
            return body
        if r < 0.7:
            return body.encode()
        if r < 0.8:
            return bytearray(body.encode())
        if r < 0.83:
            return 'AC\xe9GT'
        if r < 0.86:
            return 'ACΔGT'
        if r < 0.88:
            return memoryview(body.encode())
        return body
    calls = 0
    for trial in range(30):
        seq = [item() for _ in range(rnd.choice([0, 1, 3, 50, 700, 5000, 9000]))]
        if trial & 1:
            seq = tuple(seq)
        cap = rnd.choice([1, 10, 200, 4096, 100000, 3000000])
        threads = rnd.choice([1, 2, 3, 8, 16])
        a, b = np.full(cap + 8, 7, dtype=np.uint8), np.full(cap + 8, 7, dtype=np.uint8)
        pos = 0
        while pos <= len(seq):
            ra = restated(seq, pos, a, cap)
            rb = _kpal_gather.gather(seq, pos, b.ctypes.data, cap, threads)
            assert ra == rb and np.array_equal(a[:ra[1]], b[:rb[1]]) and np.all(b[cap:] == 7), (trial, pos, ra, rb)
            calls += 1
            nxt, _, status = ra
            if status == 0:
                break
            pos = nxt + 1 if (status == 2 or nxt == pos) else nxt
    assert calls > 500
