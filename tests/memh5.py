"""In-memory stand-in for the few h5py.File operations kPAL's profile I/O uses (kpal/klib.py:63-76,227-256):
``handle['profiles']`` (a mapping of names), ``handle['profiles/<name>'][:]``, ``create_dataset(path, data=, dtype=,
compression=)`` -> object with ``.attrs``, ``flush()``.  Test infrastructure: h5py is not installed in this image."""
import numpy as np


class Dataset(object):
    def __init__(self, data, dtype):
        self._data = np.array(data, dtype=dtype)
        self.attrs = {}

    def __getitem__(self, key):
        return self._data[key].copy()


class File(object):
    def __init__(self, name=None):
        self._profiles = {}
        if name is not None:
            self.name = name
        self.flushes = 0

    def __getitem__(self, path):
        if path == 'profiles':
            return self._profiles
        group, _, name = path.partition('/')
        if group != 'profiles':
            raise KeyError(path)
        return self._profiles[name]            # KeyError for a missing profile, as h5py

    def create_dataset(self, path, data=None, dtype=None, compression=None):
        group, _, name = path.partition('/')
        assert group == 'profiles' and name and compression == 'gzip' and dtype == 'int64'
        if name in self._profiles:
            raise ValueError('Unable to create dataset (name already exists)')   # h5py's error type
        self._profiles[name] = Dataset(data, dtype)
        return self._profiles[name]

    def flush(self):
        self.flushes += 1
