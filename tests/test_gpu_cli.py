"""The command line (kpal_amd.kmer.main, ``python -m kpal_amd``) against golden G12: the reference's
``kpal.kmer.main([...])`` for every sub-command on the tutorial files (tools/gen_golden.py g12) -- exit status,
stdout, text outputs, and for every profile file written the root attributes, the per-profile attributes and the
sha256 of the stored counts.  HDF5 files are replaced by tests/memh5.py (h5py is not in this image; the real-h5py
round trip is tools/h5_roundtrip.py, run in the build container).  Run on the GPU box: pytest -m gpu."""
import contextlib
import hashlib
import io
import json
import os
import shutil

import numpy as np
import pytest

import memh5

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def describe(handle, sort_counts):
    profiles = {}
    for name in sorted(handle['profiles']):
        ds = handle['profiles/' + name]
        counts = ds[:].astype('<i8')
        if sort_counts:
            counts = np.sort(counts)
        profiles[name] = {'attrs': dict(ds.attrs), 'sha256': hashlib.sha256(counts.tobytes()).hexdigest()}
    return profiles


def test_g12_every_subcommand(tmp_path, monkeypatch):
    from kpal_amd import files, kmer
    with open(os.path.join(GOLDEN, 'cli.json')) as fh:
        g = json.load(fh)['G12']
    for n in ('a_1', 'a_2', 'b_1', 'b_2', 'c_1', 'c_2'):
        shutil.copy(os.path.join(GOLDEN, 'tutorial', n + '.fa'), str(tmp_path))
    for name, text in g['inputs'].items():
        (tmp_path / name).write_text(text)
    store = memh5.Store()
    monkeypatch.setattr(files, 'open_profile_file', store.open)
    monkeypatch.chdir(tmp_path)
    seen = set()
    for step in g['steps']:
        argv = step['argv']
        seen.add(argv[0])
        before = set(os.listdir(str(tmp_path)))
        so, se = io.StringIO(), io.StringIO()
        status = 0
        with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
            try:
                kmer.main(argv)
            except SystemExit as e:
                status = e.code
        what = ' '.join(argv)
        assert status == step['status'], (what, se.getvalue())
        got_out = so.getvalue()
        want_out = step['stdout']
        if argv[0] == 'info':      # the producer line names the writing program
            got_out = '\n'.join(l for l in got_out.split('\n') if not l.startswith('Produced by:'))
            want_out = '\n'.join(l for l in want_out.split('\n') if not l.startswith('Produced by:'))
        assert got_out == want_out, what
        if status:
            err = se.getvalue().strip().split('\n')[-1]
            err = err.split('error: ', 1)[1] if 'error: ' in err else err
            want = step['error']
            if 'Unable to open file' in want:      # h5py's wording of the cause
                assert err.split(':')[:2] == want.split(':')[:2], what
            elif 'invalid choice' in want:         # argparse wording differs between Python versions
                assert err.startswith('argument subcommand: invalid choice'), what
            else:
                assert err == want, what
        created = sorted(set(os.listdir(str(tmp_path))) - before)
        for name, text in step['text_files'].items():
            assert (tmp_path / name).read_text() == text, (what, name)
        for name, want in step['profile_files'].items():
            handle = store.files[os.path.abspath(name)]
            assert handle.attrs['format'] == want['root']['format'] == 'kMer', what
            assert handle.attrs['version'] == want['root']['version'] == '1.0.0', what
            assert str(handle.attrs['producer']).startswith('kpal_amd'), what
            got = describe(handle, sort_counts=argv[0] == 'shuffle')
            assert sorted(got) == sorted(want['profiles']), (what, name)
            for pname, rec in want['profiles'].items():
                assert got[pname]['sha256'] == rec['sha256'], (what, name, pname)
                for key in ('length', 'total', 'non_zero'):
                    assert int(got[pname]['attrs'][key]) == rec['attrs'][key], (what, name, pname, key)
                for key in ('mean', 'median', 'std'):
                    assert abs(float(got[pname]['attrs'][key]) - rec['attrs'][key]) <= 1e-12 * max(1.0, abs(rec['attrs'][key])), (what, name, pname, key)
        assert sorted(n for n in created if n.endswith('.txt')) == sorted(step['text_files']), what
    assert seen >= {'convert', 'cat', 'count', 'merge', 'balance', 'showbalance', 'stats', 'distr', 'info', 'getcount',
                    'positive', 'scale', 'shrink', 'shuffle', 'smooth', 'distance', 'matrix'}


def test_module_entry_point(tmp_path, monkeypatch):
    """``python -m kpal_amd`` is the console entry (setup.py:46-48 of the reference installs ``kpal``)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-m', 'kpal_amd', 'count', '-h'], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0 and b'--by-record' in out.stdout and b'-k SIZE' in out.stdout
