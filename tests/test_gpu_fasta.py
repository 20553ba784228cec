"""FASTA ingest on the device (kpal_count_feed_fasta / kpal_fasta_flatten) against an independent
restatement of the tokenising kPAL delegates to Bio.SeqIO.parse (kpal/klib.py:111; Biopython's
SimpleFastaParser: skip text before the first '>', rstrip() every line, join, remove ' ' and '\\r')
and the oracle's counts.  Run on the GPU box: pytest -m gpu."""
import io
import os
import random

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    from kpal_amd import _native
    return _native.context()


def seqio_records(text):
    """(name, sequence) per record by Biopython's rules, written independently of kpal_amd.klib:
    universal newlines, lines before the first '>' skipped, name = first word of the title, every
    sequence line rstrip()-ed, ' ' and '\\r' removed from the joined record -- so a tab inside a
    line stays (and splits k-mer windows), a tab at the end of a line goes."""
    import re
    records, title, lines = [], None, []
    for line in re.split('\r\n|\r|\n', text):
        if line[:1] == '>':
            if title is not None:
                records.append((title, lines))
            title, lines = line[1:].rstrip(), []
        elif title is not None:
            lines.append(line.rstrip())
    if title is not None:
        records.append((title, lines))
    out = []
    for title, lines in records:
        words = title.split(None, 1)
        out.append((words[0] if words else '', ''.join(lines).replace(' ', '').replace('\r', '')))
    return out


def expected_flat(text):
    return b''.join(b'\n' + seq.encode('latin-1') for _, seq in seqio_records(text))


def random_fasta(rnd, n_records, max_len, eol='\n'):
    parts = []
    if rnd.random() < 0.3:
        parts.append('leading junk ACGT' + eol)
    for r in range(n_records):
        title = ''.join(rnd.choice('abc >|_0123') for _ in range(rnd.choice([0, 3, 20, 200])))
        if rnd.random() < 0.05:
            title += 'x' * 5000                      # header longer than a 4 KiB block
        parts.append('>' + title + eol)
        n = rnd.choice([0, 1, 5, rnd.randint(0, max_len)])
        seq = ''.join(rnd.choice('ACGT' * 6 + 'acgtNnRY->') for _ in range(n))
        width = rnd.choice([1, 7, 60, 70, 80, 10 ** 9])
        for i in range(0, len(seq), width):
            line = seq[i:i + width]
            if rnd.random() < 0.1:
                line = line[:len(line) // 2] + rnd.choice([' ', '\t', '  ', '\x0b', '\x0c']) + line[len(line) // 2:]
            if rnd.random() < 0.08:
                line += rnd.choice([' ', '\t', ' \t ', '\x0c'])      # trailing whitespace: stripped, k-mers join across the wrap
            if rnd.random() < 0.03:
                line = rnd.choice(['\t', ' ']) + line                # leading: a tab stays, a space goes
            parts.append(line + (eol if rnd.random() < 0.97 else eol + eol))
    text = ''.join(parts)
    if rnd.random() < 0.3:
        text = text.rstrip('\r\n')                   # no trailing newline
    return text


def test_flatten_matches_host_tokeniser(ctx):
    rnd = random.Random(5)
    cases = ['', 'no header at all\nACGT\n', '>only header', '>h\n', '>h\nACGT', '>a\nAC\nGT\n>b\n\n>c\nTT',
             '>a\r\nAC\r\nGT\r\n>b\r\nNN\r\n', 'x\n>a\nAC>GT\n>b\n A C\tG T \n',
             '>t\nAC\tGT\n', '>t\nAC\t\nGT\n', '>t\nAC \t \nGT', '>t\n\tACGT\n', '>t\nAC\x0b\x0cGT\t\t\n\t\nAC\t']
    for _ in range(60):
        cases.append(random_fasta(rnd, rnd.randint(1, 12), 3000, eol=rnd.choice(['\n', '\n', '\r\n'])))
    for _ in range(6):
        cases.append(random_fasta(rnd, rnd.randint(1, 4), 60000))   # many 4 KiB blocks, long single-line records
    for text in cases:
        got = ctx.fasta_flatten(text.encode('latin-1'))
        assert got == expected_flat(text), repr(text[:200])


def test_interior_tab_separates_windows(ctx):
    """'AC\\tGT' at k = 4: no k-mer in the reference (the tab stays in record.seq and splits the window,
    kpal/klib.py:152-156); a tab at the END of a wrapped line is stripped and the window continues."""
    from kpal_amd import klib
    assert klib.Profile.from_fasta(io.StringIO('>r\nAC\tGT\n'), 4).total == 0
    p = klib.Profile.from_fasta(io.StringIO('>r\nAC\t\nGT\n'), 4)
    assert p.total == 1 and p.counts[int('0123', 4)] == 1
    assert [r for r in klib._fasta_records(io.StringIO('>r x\nAC\tG T\t\nA\n'))] == [('r', 'AC\tGTA')]


def test_from_fasta_counts(ctx, tutorial_dir):
    from kpal_amd import klib
    rnd = random.Random(9)
    for k in (3, 8, 12):
        text = random_fasta(rnd, 30, 5000)
        p = klib.Profile.from_fasta(io.StringIO(text), k)
        seqs = [s for _, s in seqio_records(text)]
        np.testing.assert_array_equal(p.counts, oracle.from_sequences(seqs, k))
    # binary handle and a real file (60-column wrapped tutorial data)
    path = os.path.join(tutorial_dir, 'a_1.fa')
    with open(path, 'rb') as fb, open(path) as ft:
        pb = klib.Profile.from_fasta(fb, 8)
        pt = klib.Profile.from_fasta(ft, 8)
    np.testing.assert_array_equal(pb.counts, pt.counts)
    assert (int(pb.total), int(pb.non_zero)) == (18600, 16141)     # doc/tutorial.rst:44-58


def test_from_fasta_chunked_reads(ctx, monkeypatch):
    """Records straddling read chunks: the host cuts feeds back to a record boundary."""
    from kpal_amd import klib
    rnd = random.Random(13)
    text = random_fasta(rnd, 400, 900)
    seqs = [s for _, s in seqio_records(text)]
    want = oracle.from_sequences(seqs, 9)
    for chunk in (1 << 10, 4097, 1 << 16):
        monkeypatch.setattr(klib, '_FASTA_CHUNK', chunk)
        p = klib.Profile.from_fasta(io.StringIO(text), 9)
        np.testing.assert_array_equal(p.counts, want)
    # CRLF line ends, bytes handle, chunks smaller than most records, junk before the first header
    text = 'junk line\r\n' + random_fasta(rnd, 60, 6000, eol='\r\n')
    seqs = [s for _, s in seqio_records(text)]
    want = oracle.from_sequences(seqs, 7)
    for chunk in (257, 1 << 12, 1 << 20):
        monkeypatch.setattr(klib, '_FASTA_CHUNK', chunk)
        p = klib.Profile.from_fasta(io.BytesIO(text.encode('latin-1')), 7)
        np.testing.assert_array_equal(p.counts, want)


def test_large_single_record(ctx):
    """One 30 Mbase record wrapped at 60 columns: line joins across thousands of blocks."""
    buf = oracle.synth_reads(17, 0, 200000, 150, noisy=True)
    seq = np.ascontiguousarray(buf.reshape(-1, 151)[:, :150]).reshape(-1)
    lines = seq.reshape(-1, 60)
    text = b'>chr1 synthetic\n' + b'\n'.join(bytes(l) for l in lines) + b'\n'
    from kpal_amd import klib
    p = klib.Profile.from_fasta(io.BytesIO(text), 12)
    np.testing.assert_array_equal(p.counts, oracle.count_flat(seq, 12, threads=8))


def test_from_fasta_by_record_batched(ctx, monkeypatch):
    """One profile per record through the batched kernel (kpal/klib.py:114-133): counts vs the oracle
    per record, names (record name, or the 1-based index for an empty title) and prefix; records
    shorter than k, empty records and many records inside one 16-byte chunk."""
    from kpal_amd import klib
    rnd = random.Random(21)
    text = '>r1 first\nACGTNACGT\n>\nAC\n>r3\n\n>r4\nA\n>r5\nC\n>r6\nGGGTTTAAACCC\n' + random_fasta(rnd, 300, 700)
    head = '>r1 first\nACGTNACGT\n>\nAC\n>r3\n\n>r4\nA\n>r5\nC\n>r6\nGGGTTTAAACCC\n'
    short = head + random_fasta(rnd, 24, 700)            # k = 12: one 128 MiB table per record comes back over PCIe
    for k in (1, 3, 8, 12):
        text_k = short if k == 12 else text
        recs = seqio_records(text_k)
        for batch in (1 << 30, 3 * 8 * 4 ** k):          # everything in a few batches / three records per batch
            monkeypatch.setattr(klib, '_RECORD_BATCH_BYTES', batch)
            profiles = list(klib.Profile.from_fasta_by_record(io.StringIO(text_k), k, prefix='p'))
            assert len(profiles) == len(recs)
            for i, (p, (name, seq)) in enumerate(zip(profiles, recs)):
                assert p.name == 'p_' + (name or str(i + 1))
                assert p.length == k
                if i < 40 or i % 37 == 0:
                    np.testing.assert_array_equal(p.counts, oracle.from_sequences([seq], k), err_msg='record %d k=%d' % (i, k))
            total = sum(int(p.counts.sum()) for p in profiles)
            assert total == sum(int(oracle.from_sequences([seq], k).sum()) for _, seq in recs)
    # without a prefix, and the C-ABI directly with an empty record in the middle
    first = next(klib.Profile.from_fasta_by_record(io.StringIO(text), 4))
    assert first.name == 'r1'
    flat = b'ACGTACGT\n\nTTTT\n'
    t = ctx.count_records(2, flat, [0, 9, 10, 15])
    np.testing.assert_array_equal(t[0], oracle.from_sequences(['ACGTACGT'], 2))
    assert t[1].sum() == 0
    np.testing.assert_array_equal(t[2], oracle.from_sequences(['TTTT'], 2))
    with pytest.raises(ValueError):
        ctx.count_records(2, flat, [0, 9, 8, 15])


def test_from_fasta_by_record_file(tmp_path, monkeypatch):
    """from_fasta_by_record on FILES: the library reads the file itself and cuts it into pieces of whole records
    (kpal_fasta_records_file_*); with the piece size forced down to 64 .. 5000 bytes records span many pieces, pieces hold many
    records, and records longer than a piece are gathered first.  Names, order and every table against the independent
    tokeniser + the oracle; text and binary handles; CRLF; text before the first header; an untitled record; an empty file."""
    from kpal_amd import _native, klib
    rnd = random.Random(77)
    texts = ['junk before\n>r1 first\nACGTNACGT\n>\nAC\n>r3\n\n>r4\nA\n>r5\nC\n>r6\nGGGTTTAAACCC\n' + random_fasta(rnd, 120, 900),
             random_fasta(rnd, 40, 9000, eol='\r\n'), '>only\n' + 'ACGTTGCA' * 4000, 'no header at all\nACGT\n', '>a\nACGT\n>b desc\nTTGA']
    for chunk in (64, 257, 5000, 64 << 20):
        monkeypatch.setenv('KPAL_FASTA_CHUNK', str(chunk))
        ctx2 = _native.Context(_native.default_device())
        monkeypatch.delenv('KPAL_FASTA_CHUNK')
        monkeypatch.setattr(_native, 'context', lambda: ctx2)
        try:
            for t, text in enumerate(texts):
                path = tmp_path / ('rec%d_%d.fa' % (chunk, t))
                path.write_bytes(text.encode('latin-1'))
                recs = seqio_records(text)
                for k, mode in ((3, 'r'), (7, 'rb')):
                    monkeypatch.setattr(klib, '_RECORD_BATCH_BYTES', 5 * 8 * 4 ** k)
                    with open(str(path), mode) as fh:
                        profiles = list(klib.Profile.from_fasta_by_record(fh, k, prefix='f'))
                    assert [p.name for p in profiles] == ['f_' + (name or str(i + 1)) for i, (name, _) in enumerate(recs)], (chunk, t)
                    for i, (p, (_, seq)) in enumerate(zip(profiles, recs)):
                        if i < 30 or i % 11 == 0:
                            np.testing.assert_array_equal(p.counts, oracle.from_sequences([seq], k), err_msg='chunk %d text %d record %d k=%d' % (chunk, t, i, k))
                    assert sum(int(p.counts.sum()) for p in profiles) == sum(int(oracle.from_sequences([seq], k).sum()) for _, seq in recs)
            empty = tmp_path / 'empty.fa'
            empty.write_bytes(b'')
            with open(str(empty)) as fh:
                assert list(klib.Profile.from_fasta_by_record(fh, 4)) == []
        finally:
            ctx2.close()


def test_by_record_profiles_stay_in_hbm(ctx, tmp_path, monkeypatch):
    """Profiles of from_fasta_by_record keep their tables in HBM until something asks for ``counts`` (kpal/klib.py:114-133 ->
    kmer.py:683-700 without the tables crossing PCIe twice): name / length / number / total / non_zero / summary and copy()
    without a download; ProfileDistance.distance (plain, balanced, euclidean, with options) and kdistlib.distance_matrix on the
    device copies give the SAME bits as on downloaded copies; the first access to ``counts`` downloads and lets the device copy
    go -- an in-place change of that array is what every later distance sees (the reference's callers mutate counts: SURVEY 7);
    the budget of live device tables: past it batches are downloaded at once; the memory goes back when the profiles do."""
    import gc
    from kpal_amd import kdistlib, klib, metrics
    rnd = random.Random(31)
    text = random_fasta(rnd, 23, 3000)
    recs = seqio_records(text)
    k = 5
    monkeypatch.setattr(klib, '_RECORD_BATCH_BYTES', 7 * 8 * 4 ** k)         # seven records per batch: profiles of several batches
    gc.collect()
    live0 = klib._DeviceBatch.live_bytes
    profiles = list(klib.Profile.from_fasta_by_record(io.StringIO(text), k))
    assert len(profiles) == len(recs) and all(p._device_counts() is not None for p in profiles)
    assert klib._DeviceBatch.live_bytes - live0 == len(recs) * 8 * 4 ** k
    want = [oracle.from_sequences([seq], k) for _, seq in recs]
    for p, w, (name, _) in zip(profiles, want, recs):
        assert p.length == k and p.number == 4 ** k and int(p.total) == int(w.sum()) and int(p.non_zero) == int(np.count_nonzero(w))
        s = p.summary()
        assert int(s['total']) == int(w.sum()) and abs(float(s['mean']) - w.mean()) <= 1e-12 * max(1.0, w.mean()) and float(s['median']) == float(np.median(w))
    assert all(p._device_counts() is not None for p in profiles)               # nothing above downloaded a table
    host = [klib.Profile(w.copy(), name=p.name) for p, w in zip(profiles, want)]
    dists = [kdistlib.ProfileDistance(), kdistlib.ProfileDistance(do_balance=True), kdistlib.ProfileDistance(pairwise=metrics.pairwise['sum']),
             kdistlib.ProfileDistance(distance_function=metrics.euclidean), kdistlib.ProfileDistance(distance_function=metrics.cosine_similarity),
             kdistlib.ProfileDistance(do_smooth=True, summary=metrics.summary['average'], threshold=2), kdistlib.ProfileDistance(do_scale=True, down=True),
             kdistlib.ProfileDistance(do_positive=True, do_balance=True)]
    def same(a, b):
        return a == b or (a != a and b != b)              # (an empty record: nan from both)

    for d in dists:
        for i, j in ((1, 0), (8, 3), (22, 7), (15, 14)):
            assert same(d.distance(profiles[i], profiles[j]), d.distance(host[i], host[j])), (i, j)
        a, b = io.StringIO(), io.StringIO()
        kdistlib.distance_matrix(profiles, a, 10, d)
        kdistlib.distance_matrix(host, b, 10, d)
        assert a.getvalue() == b.getvalue()
    sub = profiles[7:14]                                                       # one whole batch: used where it lies
    a, b = io.StringIO(), io.StringIO()
    kdistlib.distance_matrix(sub, a, 10, dists[0])
    kdistlib.distance_matrix(host[7:14], b, 10, dists[0])
    assert a.getvalue() == b.getvalue()
    assert all(p._device_counts() is not None for p in profiles)               # distances and matrices read the device copies
    # a user-supplied callable: the reference's NumPy path on downloaded counts
    custom = kdistlib.ProfileDistance(pairwise=lambda x, y: abs(x - y) / (x + y + 2.0))
    assert same(custom.distance(profiles[2], profiles[1]), custom.distance(host[2], host[1]))
    assert profiles[2]._device_counts() is None and profiles[1]._device_counts() is None
    # copy() shares the device table; the first look at counts downloads it and the array is the truth from then on
    c = profiles[4].copy()
    assert c._device_counts() is not None and c.name == profiles[4].name
    arr = profiles[4].counts
    np.testing.assert_array_equal(arr, want[4])
    assert profiles[4]._device_counts() is None and c._device_counts() is not None
    arr[3] += 7
    changed = klib.Profile(want[4].copy())
    changed.counts[3] += 7
    assert same(dists[0].distance(profiles[4], profiles[5]), dists[0].distance(changed, host[5]))
    assert same(dists[0].distance(c, profiles[5]), dists[0].distance(host[4], host[5]))      # the copy still has the counted table
    profiles[6].counts = want[6] * 2                                           # the setter drops the device copy as well
    assert profiles[6]._device_counts() is None and int(profiles[6].total) == 2 * int(want[6].sum())
    import copy as _copy
    import pickle
    for clone in (_copy.deepcopy(profiles[12]), pickle.loads(pickle.dumps(profiles[13]))):      # the counts travel, not the device handle
        assert clone._device_counts() is None
    np.testing.assert_array_equal(_copy.deepcopy(profiles[12]).counts, want[12])
    np.testing.assert_array_equal(pickle.loads(pickle.dumps(profiles[13])).counts, want[13])
    profiles[9].balance()
    np.testing.assert_array_equal(profiles[9].counts, oracle.balance(want[9], k))
    # mixed: some on the device, some on the host -> the host path, same text
    a, b = io.StringIO(), io.StringIO()
    kdistlib.distance_matrix(profiles[:6], a, 8, dists[0])
    ref = [klib.Profile(w.copy(), name=p.name) for p, w in zip(profiles[:6], want[:6])]
    ref[4].counts[3] += 7
    kdistlib.distance_matrix(ref, b, 8, dists[0])
    assert a.getvalue() == b.getvalue()
    # saving (kmer.count --by-record) materialises; the file holds the counted table
    import memh5
    h5 = memh5.File()
    h5.create_group('profiles')
    profiles[11].name = 'rec11'
    profiles[11].save(h5)
    np.testing.assert_array_equal(h5['profiles/rec11'][:], want[11])
    assert profiles[11]._device_counts() is None
    # the budget: nothing stays on the device past it
    del profiles, sub, c, arr, p                                               # (p: the loop variable above still holds the last profile)
    gc.collect()
    assert klib._DeviceBatch.live_bytes == live0
    monkeypatch.setattr(klib, '_DEVICE_PROFILE_BYTES', 10 * 8 * 4 ** k)
    mixed = list(klib.Profile.from_fasta_by_record(io.StringIO(text), k))
    on_device = [m._device_counts() is not None for m in mixed]
    assert on_device[:7] == [True] * 7 and on_device[7:14] == [False] * 7 and sum(on_device) <= 10       # (a batch is all or nothing)
    for q, w in zip(mixed, want):
        np.testing.assert_array_equal(q.counts, w)
    del mixed, q
    gc.collect()
    assert klib._DeviceBatch.live_bytes == live0


def test_counted_profiles_stay_in_hbm(monkeypatch):
    """Profile.from_sequences / from_fasta leave the finished table in HBM as well (a device-to-device copy out of the context's
    count table): a distance between two freshly counted profiles moves no table, `total` / `non_zero` come from the device, the
    first look at ``counts`` downloads; with the budget at zero the table comes to the host at once, as before round 6."""
    import gc
    from kpal_amd import kdistlib, klib
    reads_a = oracle.synth_reads(41, 0, 3000, 150, noisy=True)
    reads_b = oracle.synth_reads(42, 0, 2500, 150, noisy=True)
    seqs_a = [bytes(r) for r in reads_a.reshape(-1, 151)[:, :150]]
    text_b = ''.join('>r%d\n%s\n' % (i, bytes(r).decode('latin-1')) for i, r in enumerate(reads_b.reshape(-1, 151)[:, :150]))
    gc.collect()
    live0 = klib._DeviceBatch.live_bytes
    for k in (5, 9, 13):
        want_a, want_b = oracle.count_flat(reads_a, k), oracle.from_sequences([s for _, s in seqio_records(text_b)], k)
        pa = klib.Profile.from_sequences(seqs_a, k, name='a')
        pb = klib.Profile.from_fasta(io.StringIO(text_b), k, name='b')
        assert pa._device_counts() is not None and pb._device_counts() is not None
        assert klib._DeviceBatch.live_bytes - live0 == 2 * 8 * 4 ** k
        assert int(pa.total) == int(want_a.sum()) and int(pb.non_zero) == int(np.count_nonzero(want_b)) and pa.number == 4 ** k
        for d, kw in ((kdistlib.ProfileDistance(), {}), (kdistlib.ProfileDistance(do_balance=True), {'do_balance': True})):
            got = d.distance(pa, pb)
            want = oracle.distance(want_a, want_b, k, **kw)
            assert abs(got - want) <= 1e-9 * abs(want), (k, kw)
        assert pa._device_counts() is not None and pb._device_counts() is not None
        # a second count on the same context does not disturb the first profile's table
        pc = klib.Profile.from_sequences(seqs_a[:100], k)
        np.testing.assert_array_equal(pa.counts, want_a)
        np.testing.assert_array_equal(pb.counts, want_b)
        assert pa._device_counts() is None
        np.testing.assert_array_equal(pc.counts, oracle.from_sequences([s.decode('latin-1') for s in seqs_a[:100]], k))
        pa.balance()
        np.testing.assert_array_equal(pa.counts, oracle.balance(want_a, k))
        del pa, pb, pc
        gc.collect()
        assert klib._DeviceBatch.live_bytes == live0
    monkeypatch.setattr(klib, '_DEVICE_PROFILE_BYTES', 0)
    ph = klib.Profile.from_sequences(seqs_a, 9)
    assert ph._device_counts() is None
    np.testing.assert_array_equal(ph.counts, oracle.count_flat(reads_a, 9))


def test_fasta_records_c_abi(ctx, tmp_path):
    """kpal_fasta_records_* through the binding: the index (header offsets into the text, record starts in the flattened stream),
    batches of any size at any record (also starting in the middle of a 16-byte chunk of the flattened text), errors, and
    two scans interleaved on one context (a generator whose index another scan has overwritten indexes its text again)."""
    from kpal_amd import klib
    text = b'junk\n>a one\nACGTAC\nGT\n>\n\n>c\nTTGACCA\n>d x y\nG\n>e\n' + b'ACGTTGCA' * 300 + b'\n>f\nCC\n'
    n, nf = ctx.fasta_records_begin(text)
    hdr, starts = ctx.fasta_records_index()
    seqs = [s for _, s in seqio_records(text.decode())]
    assert n == len(seqs) == 6 and nf == sum(len(s) + 1 for s in seqs)
    assert [text[h:h + 2] for h in hdr.tolist()] == [b'>a', b'>\n', b'>c', b'>d', b'>e', b'>f']
    assert starts.tolist() == [0] + list(np.cumsum([len(s) + 1 for s in seqs]))
    for k in (1, 2, 5):
        whole = ctx.fasta_records_count(k, 0, n)
        for r, seq in enumerate(seqs):
            np.testing.assert_array_equal(whole[r], oracle.from_sequences([seq], k), err_msg='record %d k=%d' % (r, k))
        for first in range(n):
            for cnt in range(1, n - first + 1):
                np.testing.assert_array_equal(ctx.fasta_records_count(k, first, cnt), whole[first:first + cnt])
    with pytest.raises(ValueError):
        ctx.fasta_records_count(3, 4, 3)                    # beyond the last record
    assert ctx.fasta_records_begin(b'no header here\nACGT\n') == (0, 0)
    assert ctx.fasta_records_index()[0].size == 0           # nothing indexed
    with pytest.raises(RuntimeError):
        ctx.fasta_records_file_next()                       # no file open
    # interleaved generators on the one default context: the reference's generators are independent of each other
    path = tmp_path / 'two.fa'
    path.write_bytes(b''.join(b'>r%d\nACGTACGT\n' % i for i in range(50)))
    import io
    monkey_batch = klib._RECORD_BATCH_BYTES
    klib._RECORD_BATCH_BYTES = 4 * 8 * 4 ** 3               # four records per batch: a generator has batches left to fetch when the other runs
    try:
        b = klib.Profile.from_fasta_by_record(io.BytesIO(path.read_bytes()), 3)
        assert next(b).name == 'r0'
        assert [p.name for p in klib.Profile.from_fasta_by_record(io.BytesIO(b'>y\nACG\n'), 3)] == ['y']
        rest = list(b)
        assert [p.name for p in rest] == ['r%d' % i for i in range(1, 50)]
        for p in rest:
            np.testing.assert_array_equal(p.counts, oracle.from_sequences(['ACGTACGT'], 3))
    finally:
        klib._RECORD_BATCH_BYTES = monkey_batch


def test_by_record_generators_interleaved_on_files(tmp_path, monkeypatch):
    """Two (and three) from_fasta_by_record generators over DIFFERENT files advanced in turns -- zip(), one nested in the other,
    one abandoned half-way -- on the process-wide context, with pieces of 300 bytes (every file is many pieces) and two records
    per batch: the scan state of a file lives in the generator, a piece another scan has overwritten is indexed again.  Names,
    order and every table against the tokeniser + the oracle (the reference's generators are independent: klib.py:114-133)."""
    from kpal_amd import _native, klib
    rnd = random.Random(5)
    texts = [random_fasta(rnd, 40, 500), 'junk\n' + random_fasta(rnd, 25, 1200, eol='\r\n'), random_fasta(rnd, 60, 200)]
    paths = []
    for t, text in enumerate(texts):
        path = tmp_path / ('inter%d.fa' % t)
        path.write_bytes(text.encode('latin-1'))
        paths.append(str(path))
    recs = [seqio_records(t) for t in texts]
    k = 4
    monkeypatch.setenv('KPAL_FASTA_CHUNK', '300')
    ctx2 = _native.Context(_native.default_device())
    monkeypatch.delenv('KPAL_FASTA_CHUNK')
    monkeypatch.setattr(_native, 'context', lambda: ctx2)
    monkeypatch.setattr(klib, '_RECORD_BATCH_BYTES', 2 * 8 * 4 ** k)

    def check(profile, t, i):
        name, seq = recs[t][i]
        assert profile.name == (name or str(i + 1)), (t, i)
        np.testing.assert_array_equal(profile.counts, oracle.from_sequences([seq], k), err_msg='file %d record %d' % (t, i))

    try:
        handles = [open(p_, 'rb') for p_ in paths]
        try:
            gens = [klib.Profile.from_fasta_by_record(h, k) for h in handles]
            seen = [0, 0, 0]
            for round_ in range(max(len(r) for r in recs)):
                for t, g in enumerate(gens):
                    if seen[t] < len(recs[t]):
                        check(next(g), t, seen[t])
                        seen[t] += 1
            for g in gens:
                assert list(g) == []
            assert seen == [len(r) for r in recs]
        finally:
            for h in handles:
                h.close()
        # nested: for every fifth record of file 0 a whole scan of file 2; file 1 abandoned after three records
        with open(paths[0]) as h0, open(paths[1], 'rb') as h1:
            abandoned = klib.Profile.from_fasta_by_record(h1, k)
            for i in range(3):
                check(next(abandoned), 1, i)
            for i, p0 in enumerate(klib.Profile.from_fasta_by_record(h0, k)):
                check(p0, 0, i)
                if i % 5 == 0:
                    with open(paths[2]) as h2:
                        for j, p2 in enumerate(klib.Profile.from_fasta_by_record(h2, k)):
                            check(p2, 2, j)
                        assert j == len(recs[2]) - 1
            assert i == len(recs[0]) - 1
            check(next(abandoned), 1, 3)
    finally:
        ctx2.close()


def test_chunk_seams_everywhere(tmp_path):
    """The pipelined ingest (kpal_count_feed_fasta / _file: chunks cut ANYWHERE, flattened with the state the previous chunk
    left -- line start / inside a header / inside a sequence line --, k-mer windows carried across the seams by the saved tail)
    with the chunk size forced down to 16 .. 5000 bytes (KPAL_FASTA_CHUNK), so that seams fall inside headers, between '\\r'
    and '\\n', inside runs of blanks, before the first header, and chunks flatten to fewer than k - 1 bytes: flattened stream
    and counts against the host tokeniser and the oracle, from memory and from a file (any begin / end / prefix)."""
    from kpal_amd import _native
    rnd = random.Random(31)
    texts = ['no header at all\nACGT\n', '>only header', '>h\nACGT', 'junk\njunk2\n>a desc\nAC\nGT\n>b\n\n>c\nTT',
             '>a\r\nAC\r\nGT\r\n>b\r\nNN\r\nACGTACGTAC\r\n', '>t\nAC \t \nGT\tA  \t\nACGT\n', '>' + 'h' * 300 + '\nACGTACGTACGTACGT\n' * 5,
             'junk\r>a x\rACGTAC\rGTTGCA\r>b\r\rTTGACC\r', '>only\n' + 'ACGT' * 50 + ' \t' * 40 + '\n' + 'TTGCA' * 30 + '\t' * 9 + 'ACG\n',
             # runs of blanks longer than a chunk (several whole chunks of nothing but blanks): trailing their line, inside it, at the
             # end of the text with and without an end of line -- what they are is decided by the first byte BEHIND the chunk
             '>long\nACGTACGTACGT' + '\t ' * 3000 + '\nTTGACCAGT' + '\t' * 5500 + 'ACGTTGCAACGT\n>next\nAC' + ' \x0b' * 2600,
             '>tail\nACGTACGTACGTA' + '\t' * 700 + '\r\n' + '\x0c' * 9000 + '\r\nGGGTTTACGTAC\t' + ' ' * 6000 + '\tACGTACGTTT' + '\t' * 5001 + '\n']
    for _ in range(10):
        texts.append(random_fasta(rnd, rnd.randint(1, 12), 3000, eol=rnd.choice(['\n', '\n', '\r\n'])))
    texts.append(random_fasta(rnd, 3, 60000))
    for chunk in (16, 17, 64, 257, 5000):            # (the host side of the seams: tests/native/fasta_host_check.cpp, every size from 16 bytes)
        os.environ['KPAL_FASTA_CHUNK'] = str(chunk)
        try:
            c = _native.Context(_native.default_device())
        finally:
            del os.environ['KPAL_FASTA_CHUNK']
        for t, text in enumerate(texts):
            if chunk < 64 and len(text) > 20000:
                continue
            raw = text.encode('latin-1')
            assert c.fasta_flatten(raw) == expected_flat(text), (chunk, repr(text[:100]))
            seqs = [s for _, s in seqio_records(text)]
            path = tmp_path / ('c%d_%d.fa' % (chunk, t))
            path.write_bytes(raw)
            for k in (1, 3, 9) + ((12,) if t < 8 and chunk in (17, 257) else ()):
                want = oracle.from_sequences(seqs, k)
                c.count_begin(k)
                c.count_feed_fasta(raw)
                np.testing.assert_array_equal(c.count_finish(), want, err_msg='memory chunk=%d k=%d %r' % (chunk, k, text[:80]))
                c.count_begin(k)
                c.count_feed_fasta_file(str(path))
                np.testing.assert_array_equal(c.count_finish(), want, err_msg='file chunk=%d k=%d %r' % (chunk, k, text[:80]))
        c.close()


def test_feed_fasta_file_ranges_and_errors(ctx, tmp_path):
    """kpal_count_feed_fasta_file: byte ranges, a prefix, several feeds into one count (windows never span feeds), errors."""
    text = b'>a\nACGTACGTAC\nGGTTAACC\n>b\nTTTTACGT\n'
    path = tmp_path / 'r.fa'
    path.write_bytes(text)
    k = 4
    whole = oracle.from_sequences(['ACGTACGTACGGTTAACC', 'TTTTACGT'], k)
    ctx.count_begin(k)
    ctx.count_feed_fasta_file(str(path))
    np.testing.assert_array_equal(ctx.count_finish(), whole)
    # the second record alone; the first alone through an explicit end
    at = text.index(b'>b')
    ctx.count_begin(k)
    ctx.count_feed_fasta_file(str(path), at)
    np.testing.assert_array_equal(ctx.count_finish(), oracle.from_sequences(['TTTTACGT'], k))
    ctx.count_begin(k)
    ctx.count_feed_fasta_file(str(path), 0, at)
    np.testing.assert_array_equal(ctx.count_finish(), oracle.from_sequences(['ACGTACGTACGGTTAACC'], k))
    # a cut inside record a, after its first line: the right-hand range with the k - 1 bases before the cut as prefix
    cut = text.index(b'GGTT')
    ctx.count_begin(k)
    ctx.count_feed_fasta_file(str(path), 0, cut)
    ctx.count_feed_fasta_file(str(path), cut, 0, b'>\n' + b'TAC')
    np.testing.assert_array_equal(ctx.count_finish(), whole)
    # without the prefix the range starts with text before its first header: ignored up to '>b'
    ctx.count_begin(k)
    ctx.count_feed_fasta_file(str(path), cut)
    np.testing.assert_array_equal(ctx.count_finish(), oracle.from_sequences(['TTTTACGT'], k))
    with pytest.raises(OSError):
        ctx.count_feed_fasta_file(str(tmp_path / 'missing.fa'))
    with pytest.raises(OSError):
        ctx.count_feed_fasta_file(str(tmp_path))
    with pytest.raises(ValueError):
        ctx.count_feed_fasta_file(str(path), 10, 5)
    with pytest.raises(ValueError):
        ctx.count_feed_fasta_file(str(path), 0, len(text) + 1)
    ctx.count_finish()


def test_from_fasta_file_handles_take_the_library_reader(ctx, tmp_path, monkeypatch):
    """Profile.from_fasta on ordinary file handles (text or binary, what `kpal count` opens) goes through
    kpal_count_feed_fasta_file -- no text passes through Python; handles whose bytes are not their text (StringIO, gzip,
    a handle somebody has read from in text mode, UTF-16) take the chunked reads.  Same counts either way."""
    import gzip
    from kpal_amd import _native, klib
    rnd = random.Random(77)
    text = 'junk\n' + random_fasta(rnd, 200, 2000)
    seqs = [s for _, s in seqio_records(text)]
    want = oracle.from_sequences(seqs, 10)
    path = tmp_path / 'h.fa'
    path.write_bytes(text.encode('latin-1'))
    calls = []
    real = _native.Context.count_feed_fasta_file
    monkeypatch.setattr(_native.Context, 'count_feed_fasta_file', lambda self, *a, **kw: (calls.append(a), real(self, *a, **kw))[1])
    for opener in (lambda: open(path), lambda: open(path, 'rb'), lambda: open(path, encoding='latin-1'), lambda: open(path, 'rb', buffering=0)):
        with opener() as fh:
            p = klib.Profile.from_fasta(fh, 10, name='x')
            assert fh.read(1) in ('', b'')             # consumed
        np.testing.assert_array_equal(p.counts, want)
    assert len(calls) == 4
    # a binary handle positioned after the junk line: the library starts there
    with open(path, 'rb') as fh:
        fh.readline()
        p = klib.Profile.from_fasta(fh, 10)
    np.testing.assert_array_equal(p.counts, want)
    assert len(calls) == 5 and calls[-1][1] == 5
    # not plain files
    with gzip.open(str(path) + '.gz', 'wt', encoding='latin-1') as fh:
        fh.write(text)
    with open(str(path) + '.u16', 'w', encoding='utf-16') as fh:
        fh.write(text.encode('latin-1').decode('latin-1'))
    for opener in (lambda: io.StringIO(text), lambda: gzip.open(str(path) + '.gz', 'rt', encoding='latin-1'),
                   lambda: gzip.open(str(path) + '.gz', 'rb'), lambda: open(str(path) + '.u16', encoding='utf-16')):
        with opener() as fh:
            p = klib.Profile.from_fasta(fh, 10)
        np.testing.assert_array_equal(p.counts, want)
    with open(path) as fh:
        fh.readline()                                   # a text handle that has been read from: its position is opaque
        p = klib.Profile.from_fasta(fh, 10)
    np.testing.assert_array_equal(p.counts, want)
    assert len(calls) == 5


def test_multi_chunk_file_and_shards(ctx, tmp_path):
    """A 200 MB FASTA file (several 64 MiB chunks in flight; one 150 Mbase record in 60-column lines after a few thousand short
    ones): Profile.from_fasta against the oracle at k = 12; then the same file cut by kpal_amd.dist.fasta_shards for 1, 3 and 8
    ranks -- every shard counted through kpal_count_feed_fasta_file, the tables added (on one GPU: into one count) -- equals
    the whole-file profile: record-boundary cuts AND cuts inside the giant record with their halo prefixes."""
    from kpal_amd import dist, klib
    buf = oracle.synth_reads(23, 0, 1_000_000, 150, noisy=True)
    seq = np.ascontiguousarray(buf.reshape(-1, 151)[:, :150]).reshape(-1)
    short = oracle.synth_reads(24, 0, 5000, 150, noisy=True)
    path = tmp_path / 'big.fa'
    with open(path, 'wb') as fh:
        for r in short.reshape(-1, 151):
            fh.write(b'>read\n' + bytes(r[:150]) + b'\n')
        fh.write(b'>chr1 synthetic\n')
        lines = np.empty((seq.size // 60, 61), dtype=np.uint8)
        lines[:, :60] = seq.reshape(-1, 60)
        lines[:, 60] = 10
        fh.write(lines.tobytes())
        fh.write(b'>last\nACGTACGTACGTTTGA\n')
    k = 12
    want = oracle.count_flat(short, k, threads=8) + oracle.count_flat(seq, k, threads=16) + oracle.from_sequences(['ACGTACGTACGTTTGA'], k)
    with open(path) as fh:
        p = klib.Profile.from_fasta(fh, k)
    np.testing.assert_array_equal(p.counts, want)
    for world in (1, 3, 8):
        shards = dist.fasta_shards(str(path), world, k)
        sizes = [sum(s.end - s.begin for s in segs) for segs in shards]
        assert sum(sizes) == os.path.getsize(path) and max(sizes) - min(sizes) < 1 << 20
        if world > 1:
            assert any(seg.prefix for segs in shards for seg in segs)
        ctx.count_begin(k)
        for segs in shards:
            for seg in segs:
                ctx.count_feed_fasta_file(seg.path, seg.begin, seg.end, seg.prefix)
        np.testing.assert_array_equal(ctx.count_finish(), want, err_msg='world %d' % world)
    # per-shard tables, added on the host (what the reduce does)
    acc = np.zeros(4 ** k, dtype=np.int64)
    for segs in dist.fasta_shards(str(path), 4, k):
        dist.count_fasta_sharded(ctx, k, segs)
        acc += ctx.count_finish()
    np.testing.assert_array_equal(acc, want)


def test_sharded_fasta_count_through_rccl_world_1(tmp_path, tutorial_dir):
    """The multi-GPU FASTA entry as a user starts it -- python -m torch.distributed.run --nproc-per-node 1 -m kpal_amd.dist count
    -k 8 a_1.fa a_2.fa out.k8 (world size 1 here: the process group and the reduce path are RCCL's) -- writes ONE profile over
    both files: equal to the merge of the two per-file profiles (doc/tutorial.rst:94-95).  (h5py is not in this image: the
    in-memory HDF5 stand-in of the CLI tests takes the file's place.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ('import sys, os, io\n'
            'sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))\n'
            'import numpy as np, memh5\n'
            'from kpal_amd import files, dist, klib\n'
            'store = memh5.Store(); files.open_profile_file = store.open\n'
            'rc = dist.main(["count", "-k", "8", %r, %r, "out.k8"])\n'
            'assert rc == 0\n'
            'h = store.open("out.k8", "r")\n'
            'got = klib.Profile.from_file(h)\n'
            'a = klib.Profile.from_fasta(open(%r), 8); b = klib.Profile.from_fasta(open(%r), 8); a.merge(b)\n'
            'assert got.name == "a_1" and np.array_equal(got.counts, a.counts) and int(got.total) == int(a.total)\n'
            'print("DIST_COUNT_OK", int(got.total))\n') % (root, root, os.path.join(tutorial_dir, 'a_1.fa'), os.path.join(tutorial_dir, 'a_2.fa'),
                                                              os.path.join(tutorial_dir, 'a_1.fa'), os.path.join(tutorial_dir, 'a_2.fa'))
    script = tmp_path / 'dist_count.py'
    script.write_text(code)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                        '--master-port', '29611', str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and 'DIST_COUNT_OK' in out, out[-3000:]


def test_sharded_fasta_count_three_ranks_on_one_gpu(tmp_path, tutorial_dir):
    """The same entry with THREE ranks: they share the one GPU of this box (--device 0) and merge over gloo (--backend gloo; RCCL
    refuses several ranks on one device).  Every rank counts its byte range of the two files (cuts inside records, k - 1-base halo),
    rank 0 writes the one profile: equal to the merge of the per-file profiles."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a1, a2 = os.path.join(tutorial_dir, 'a_1.fa'), os.path.join(tutorial_dir, 'a_2.fa')
    code = ('import sys, os, io\n'
            'sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))\n'
            'import numpy as np, memh5\n'
            'from kpal_amd import files, dist, klib\n'
            'store = memh5.Store(); files.open_profile_file = store.open\n'
            'out = %r\n'
            'rc = dist.main(["count", "-k", "8", "--backend", "gloo", "--device", "0", %r, %r, out])\n'
            'assert rc == 0\n'
            'if os.environ["RANK"] == "0":\n'
            '    got = klib.Profile.from_file(store.open(out, "r"))\n'
            '    a = klib.Profile.from_fasta(open(%r), 8); b = klib.Profile.from_fasta(open(%r), 8); a.merge(b)\n'
            '    assert got.name == "a_1" and np.array_equal(got.counts, a.counts) and int(got.total) == int(a.total)\n'
            '    print("DIST_COUNT_OK", int(got.total))\n'
            'else:\n'
            '    assert not store.files\n') % (root, root, str(tmp_path / 'out3.k8'), a1, a2, a1, a2)
    script = tmp_path / 'dist_count3.py'
    script.write_text(code)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3', '--master-addr', '127.0.0.1',
                        '--master-port', '29613', str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = p.stdout.decode()
    assert p.returncode == 0 and 'DIST_COUNT_OK' in out, out[-3000:]


@pytest.mark.parametrize('env', [{'KPAL_READ_THREADS': '1'}, {'KPAL_READ_PIN': '0', 'KPAL_READ_THREADS': '3'}])
def test_ingest_with_other_pool_settings(tmp_path, env):
    """The host copy pool (csrc/host_pool.hpp) is sized and bound when it is first used, once per process: a child process per
    setting -- a pool of one (every copy on the calling thread), three unbound threads -- counts a multi-chunk FASTA file and a
    large host feed and must reproduce the default's tables."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    buf = oracle.synth_reads(29, 0, 600000, 150, noisy=True)                     # 90 MB: two staging pieces, several preads each
    path = tmp_path / 'p.fa'
    with open(path, 'wb') as fh:
        fh.write(b'>r\n')
        fh.write(buf.tobytes())
    want = oracle.count_flat(buf, 11, threads=8)                                 # as a host feed: every line a sequence of its own
    joined = np.ascontiguousarray(buf.reshape(-1, 151)[:, :150]).reshape(-1)     # as ONE FASTA record: the lines are joined
    np.save(tmp_path / 'want.npy', want)
    np.save(tmp_path / 'want_record.npy', oracle.count_flat(joined, 11, threads=8))
    code = ('import sys, numpy as np\n'
            'sys.path.insert(0, %r)\n'
            'from kpal_amd import klib, _native\n'
            'with open(%r) as fh:\n'
            '    p = klib.Profile.from_fasta(fh, 11)\n'
            'assert np.array_equal(p.counts, np.load(%r)), "file"\n'
            'ctx = _native.context()\n'
            'raw = np.fromfile(%r, dtype=np.uint8)[3:]\n'
            'assert np.array_equal(ctx.count_bytes(11, raw), np.load(%r)), "host feed"\n'
            'print("POOL_OK")\n') % (root, str(path), str(tmp_path / 'want_record.npy'), str(path), str(tmp_path / 'want.npy'))
    p = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b'POOL_OK' in p.stdout, p.stdout.decode()[-2000:]
