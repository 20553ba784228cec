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
