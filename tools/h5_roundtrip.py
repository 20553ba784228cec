#!/opt/conda/bin/python3.9
"""HDF5 interoperability proof with REAL h5py (build container only: the system interpreter and the GPU box
have no h5py, the conda interpreter has it together with what the unmodified reference needs):

    PYTHONPATH=tools/oracle_stubs:/root/reference /opt/conda/bin/python3.9 tools/h5_roundtrip.py

  1. files written by kpal_amd (files.ProfileFileType('w') + klib.Profile.save) are opened by the reference's
     ProfileFileType('r') (format / version check, kpal/__init__.py:84-111), loaded by its Profile.from_file and
     printed by its kmer.info;
  2. files written by the reference are opened, loaded and printed by kpal_amd;
  3. dataset layout (doc/fileformat.rst:23-46): /profiles/<name>, int64, gzip, the six attributes with the same
     values and HDF5 types, root attributes format / version / producer.

There is no GPU here, and kpal_amd has no CPU fallback for the five summary attributes (they come from one
kpal_stats pass on the device).  This script therefore substitutes Profile._device_stats by NumPy -- for this
check only: what is under test is the FILE logic, not the arithmetic (that is pinned by golden G10 / G11 / G12 on
the GPU box)."""
from __future__ import print_function

import io
import os
import sys
import tempfile

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import kpal                                   # the reference  # noqa: E402
from kpal import klib as ref_klib, kmer as ref_kmer   # noqa: E402
from kpal_amd import files, klib, kmer          # noqa: E402


class _Stats(object):
    def __init__(self, c):
        self.total, self.non_zero = int(c.sum()), int(np.count_nonzero(c))
        self.mean, self.median, self.std = float(c.mean()), float(np.median(c)), float(c.std())


klib.Profile._device_stats = lambda self: _Stats(np.asarray(self.counts))   # see the docstring


def dataset_facts(path):
    with h5py.File(path, 'r') as f:
        out = {'root': dict((k, (v.decode() if isinstance(v, bytes) else str(v))) for k, v in f.attrs.items())}
        for name in sorted(f['profiles']):
            ds = f['profiles/' + name]
            out[name] = {'dtype': str(ds.dtype), 'compression': ds.compression, 'shape': ds.shape,
                         'attrs': dict((k, (type(v).__name__, v.item() if hasattr(v, 'item') else v)) for k, v in sorted(ds.attrs.items()))}
        return out


def main():
    tmp = tempfile.mkdtemp()
    rs = np.random.RandomState(5)
    profiles = [('alpha', rs.poisson(3.0, 4 ** 5).astype('int64')), ('beta', rs.poisson(40.0, 4 ** 5).astype('int64')),
                (None, np.arange(4 ** 3, dtype='int64'))]
    ours, theirs = os.path.join(tmp, 'ours.k5'), os.path.join(tmp, 'theirs.k5')
    h = files.ProfileFileType('w')(ours)
    for name, counts in profiles:
        klib.Profile(counts.copy(), name).save(h)
    h.close()
    h = kpal.ProfileFileType('w')(theirs)
    for name, counts in profiles:
        ref_klib.Profile(counts.copy(), name).save(h)
    h.close()
    fo, ft = dataset_facts(ours), dataset_facts(theirs)
    assert fo['root']['format'] == ft['root']['format'] == 'kMer'
    assert fo['root']['version'] == ft['root']['version'] == '1.0.0'
    assert sorted(fo) == sorted(ft) == ['1', 'alpha', 'beta', 'root'], sorted(fo)
    for name in ('1', 'alpha', 'beta'):
        assert fo[name]['dtype'] == ft[name]['dtype'] == 'int64'
        assert fo[name]['compression'] == ft[name]['compression'] == 'gzip'
        assert fo[name]['shape'] == ft[name]['shape']
        assert sorted(fo[name]['attrs']) == sorted(ft[name]['attrs']) == ['length', 'mean', 'median', 'non_zero', 'std', 'total']
        for key in fo[name]['attrs']:
            (to, vo), (tt, vt) = fo[name]['attrs'][key], ft[name]['attrs'][key]
            assert vo == vt, (name, key, vo, vt)
            print('   attr %-8s %-6s ours %-8s reference %-8s value %r' % (key, name, to, tt, vo))
    # 1. the reference reads ours
    h = kpal.ProfileFileType('r')(ours)
    for name, counts in profiles:
        p = ref_klib.Profile.from_file(h, name=name or '1')
        assert np.array_equal(p.counts, counts) and p.counts.dtype == np.int64
    buf = io.StringIO()
    ref_kmer.info(h, buf)
    ref_info_of_ours = buf.getvalue()
    h.close()
    # 2. we read theirs
    h = files.ProfileFileType('r')(theirs)
    for name, counts in profiles:
        p = klib.Profile.from_file(h, name=name or '1')
        assert np.array_equal(p.counts, counts) and p.counts.dtype == np.int64
    buf = io.StringIO()
    kmer.info(h, buf)
    our_info_of_theirs = buf.getvalue()
    h.close()
    strip = lambda text: '\n'.join(l for l in text.split('\n') if not l.startswith('Produced by:'))   # noqa: E731
    assert strip(ref_info_of_ours) == strip(our_info_of_theirs)
    # 3. both refuse to overwrite, both refuse foreign files
    for opener in (files.ProfileFileType('w'), kpal.ProfileFileType('w')):
        try:
            opener(ours)
            raise SystemExit('overwrote an existing file')
        except Exception as error:   # argparse.ArgumentTypeError
            assert 'file exists' in str(error)
    with h5py.File(os.path.join(tmp, 'foreign.h5'), 'w') as f:
        f.attrs['format'] = 'other'
    for opener in (files.ProfileFileType('r'), kpal.ProfileFileType('r')):
        try:
            opener(os.path.join(tmp, 'foreign.h5'))
            raise SystemExit('accepted a foreign file')
        except Exception as error:
            assert 'not a k-mer profile file' in str(error)
    print('reference kmer.info on a file written by kpal_amd:')
    print(ref_info_of_ours)
    print('HDF5 ROUND TRIP OK: h5py %s, numpy %s, reference kPAL %s <-> kpal_amd (%s)' % (
        h5py.__version__, np.__version__, kpal.__version__, fo['root']['producer']))


if __name__ == '__main__':
    main()
