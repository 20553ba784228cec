"""In-memory stand-in for the few h5py.File operations kPAL's profile I/O uses (kpal/klib.py:63-76,227-256,
kpal/__init__.py:83-111): ``handle['profiles']`` (a mapping of names), ``handle['profiles/<name>'][:]``,
``create_dataset(path, data=, dtype=, compression=)`` -> object with ``.attrs``, root ``attrs``, ``create_group``,
``flush()``; and ``Store``, which stands in for ``h5py.File(path, mode)`` behind ``kpal_amd.files.open_profile_file``.
Test infrastructure: h5py is not installed in this image."""
import os

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
        self.attrs = {}
        if name is not None:
            self.name = name
        self.flushes = 0

    def create_group(self, name):
        assert name == 'profiles'

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

    def close(self):
        self.flushes += 1


class Store(object):
    """``open(path, mode)`` in place of ``h5py.File``: files written live in memory, an empty placeholder is
    created on disk so that the overwrite protection (``os.path.exists``) sees them."""

    def __init__(self):
        self.files = {}

    def open(self, path, mode):
        key = os.path.abspath(path)
        if 'w' in mode:
            handle = File(name=path)
            self.files[key] = handle
            open(path, 'wb').close()
            return handle
        if key in self.files:
            return self.files[key]
        if not os.path.exists(path):
            raise IOError("[Errno 2] Unable to open file (unable to open file: name = '%s', errno = 2)" % path)
        raise IOError('Unable to open file (file signature not found)')
