"""The callers of SURVEY.md section 8 row a14 (kpal/kmer.py:112-271,541-700) through kpal_amd.kmer against
outputs of the reference's own functions on real HDF5 files (golden G11, tools/gen_golden.py g11): stored counts
bit-exact (sha256), integer attributes exact, float attributes within 1e-9, text outputs identical at the
printed precisions.  Profile files are tests/memh5.py handles (h5py is not in this image).  pytest -m gpu."""
import hashlib
import io
import os

import numpy as np
import pytest

import memh5

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def g11(request):
    import json
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'callers.json')) as fh:
        return json.load(fh)['G11']


def check_profiles(handle, want):
    assert sorted(handle['profiles']) == sorted(want)
    for name, rec in want.items():
        ds = handle['profiles/' + name]
        assert hashlib.sha256(ds[:].astype('<i8').tobytes()).hexdigest() == rec['sha256'], name
        for key, value in rec['attrs'].items():
            got = ds.attrs[key]
            if isinstance(value, float):
                assert abs(float(got) - value) <= 1e-9 * abs(value) + 1e-300, (name, key, got, value)
            else:
                assert int(got) == value, (name, key, got, value)


def same_numbers(text, want, precision):
    """Identical layout and names; numbers equal as printed, or off by one unit in the last of
    >= 8 printed decimals (an fp64 sum in a different order may round the other way)."""
    if text == want:
        return True
    a, b = text.split('\n'), want.split('\n')
    if len(a) != len(b):
        return False
    for la, lb in zip(a, b):
        ta, tb = la.split(' '), lb.split(' ')
        if len(ta) != len(tb):
            return False
        for x, y in zip(ta, tb):
            if x == y:
                continue
            if precision < 8 or abs(float(x) - float(y)) > 1.5 * 10.0 ** -precision:
                return False
    return True


@pytest.fixture(scope='module')
def counted(g11, tutorial_dir):
    from kpal_amd import kmer
    handles = [open(os.path.join(tutorial_dir, n + '.fa')) for n in g11['files']]
    out = memh5.File()
    kmer.count(handles, out, 8)                       # names come from the file names
    for h in handles:
        h.close()
    return out


def test_count_names_attrs_and_counts(g11, counted, tutorial_dir):
    from kpal_amd import kmer
    check_profiles(counted, g11['count_k8'])
    assert counted.flushes == len(g11['files'])
    handles = [open(os.path.join(tutorial_dir, n + '.fa'), 'rb') for n in g11['files'][:2]]   # binary handles too
    named = memh5.File()
    kmer.count(handles, named, 5, names=['x', 'y'])
    check_profiles(named, g11['count_k5_named'])
    with pytest.raises(ValueError, match='number of profile names does not match'):
        kmer.count(handles, memh5.File(), 5, names=['x'])
    # nameless handles: numbered from 1
    out = memh5.File()
    kmer.count([io.StringIO('>a\nACGT\n'), io.StringIO('>b\nTTTT\n')], out, 2)
    assert sorted(out['profiles']) == ['1', '2']


def test_count_by_record(g11):
    from kpal_amd import kmer
    one = memh5.File()
    kmer.count([io.StringIO(g11['by_record_one_file']['input'])], one, 4, by_record=True)
    check_profiles(one, g11['by_record_one_file']['profiles'])
    two = memh5.File()
    kmer.count([io.StringIO(t) for t in g11['by_record_two_files']['inputs']], two, 3, names=['p', 'q'], by_record=True)
    check_profiles(two, g11['by_record_two_files']['profiles'])


def test_merge_and_balance(g11, counted):
    from kpal_amd import kmer
    for merger, want in g11['merge'].items():
        out = memh5.File()
        kmer.merge(counted, counted, out, names_left=['a_1', 'b_1'], names_right=['a_2', 'b_2'], merger=merger)
        check_profiles(out, want)
    out = memh5.File()
    kmer.merge(counted, counted, out, names_left=['c_1'], names_right=['c_1'])
    check_profiles(out, g11['merge_same_name'])
    out = memh5.File()
    kmer.merge(counted, counted, out, names_left=['c_1'], names_right=['c_2'], custom_merger='np.maximum(left, right)')
    check_profiles(out, g11['merge_custom'])
    with pytest.raises(ValueError, match='left and right profile names do not match'):
        kmer.merge(counted, counted, memh5.File(), names_left=['a_1'], names_right=['a_1', 'a_2'])
    out = memh5.File()
    kmer.balance(counted, out, names=['a_1', 'c_2'])
    check_profiles(out, g11['balance'])
    # different k in the two files
    other = memh5.File()
    kmer.count([io.StringIO('>s\nACGTACGT\n')], other, 3, names=['a_1'])
    with pytest.raises(ValueError, match='k-mer lengths of the files differ'):
        kmer.merge(counted, other, memh5.File(), names_left=['a_1'], names_right=['a_1'])


def test_showbalance_and_stats_text(g11, counted):
    from kpal_amd import kmer
    for key, fn, kw, precision in (('get_balance_p10', kmer.get_balance, {}, 10), ('get_balance_p3', kmer.get_balance, {'precision': 3}, 3),
                                   ('get_stats_p10', kmer.get_stats, {}, 10),
                                   ('get_stats_p4', kmer.get_stats, {'precision': 4, 'names': ['b_2', 'a_1']}, 4)):
        buf = io.StringIO()
        fn(counted, buf, **kw)
        assert same_numbers(buf.getvalue(), g11[key], precision), (key, buf.getvalue(), g11[key])


def test_distance_and_matrix_text(g11, counted):
    from kpal_amd import kmer, klib
    left, right = memh5.File(), memh5.File()
    for n in 'abc':
        klib.Profile.from_file(counted, n + '_1').save(left, name=n)
        klib.Profile.from_file(counted, n + '_2').save(right, name=n)
    assert len(g11['distance']) >= 12 and len(g11['matrix']) >= 5
    for case in g11['distance']:
        buf = io.StringIO()
        kmer.distance(left, right, buf, **case['kwargs'])
        assert same_numbers(buf.getvalue(), case['text'], case['kwargs'].get('precision', 10)), (case, buf.getvalue())
    for case in g11['matrix']:
        buf = io.StringIO()
        kmer.distance_matrix(counted, buf, **case['kwargs'])
        assert same_numbers(buf.getvalue(), case['text'], case['kwargs'].get('precision', 10)), (case, buf.getvalue())
    with pytest.raises(ValueError, match='at least two'):
        kmer.distance_matrix(counted, io.StringIO(), names=['a_1'])
    with pytest.raises(ValueError, match='left and right profile names do not match'):
        kmer.distance(left, right, io.StringIO(), names_left=['a'], names_right=['a', 'b'])
