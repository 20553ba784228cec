"""Command-line file types of the kPAL front end (kpal/__init__.py:44-111): ``FileType`` (text files that are
never overwritten, ``-`` for the standard streams) and ``ProfileFileType`` (HDF5 k-mer profile files: header
written on creation, format and version checked on opening), plus the small helpers ``kpal.kmer.main`` uses.

File format 1.0.0 (doc/fileformat.rst:23-46): root attributes ``format = 'kMer'``, ``version = '1.0.0'``,
``producer``; group ``profiles``; one gzip-compressed int64 dataset per profile with the attributes ``length``,
``total``, ``non_zero``, ``mean``, ``median``, ``std`` (written by ``klib.Profile.save``).  Files written here are
read by stock kPAL and vice versa (tools/h5_roundtrip.py proves it with real h5py in the build container).

``h5py`` is imported lazily, on the first profile file opened: the counting / distance library itself never
needs it.
"""
from __future__ import print_function

import argparse
import io
import os
import sys

from . import __version__

USAGE = ('kpal_amd (MI355X-native drop-in for kPAL, the k-mer profile analysis library)',
         'Analysis toolkit and programming library for k-mer profiles; counting and distances run as\n'
         'hand-written HIP kernels (gfx950).  Command line and file format follow kPAL 2.1.')
FORMAT_NAME = 'kMer'
FORMAT_VERSION = '1.0.0'
PRODUCER = 'kpal_amd %s (kPAL-compatible)' % __version__


def _version_tuple(text):
    """``major.minor.patch`` of a semantic version string (pre-release / build suffixes ignored)."""
    if isinstance(text, bytes):
        text = text.decode('ascii', 'replace')
    core = str(text).split('-', 1)[0].split('+', 1)[0]
    parts = core.split('.')
    if len(parts) != 3:
        raise ValueError('invalid version %r' % (text,))
    return tuple(int(p) for p in parts)


def format_version_accepted(text):
    """The reference accepts ``>=1.0.0,<2.0.0`` (kpal/__init__.py:41)."""
    return (1, 0, 0) <= _version_tuple(text) < (2, 0, 0)


def open_profile_file(path, mode):
    """The one place HDF5 files are opened (tests substitute an in-memory store here)."""
    try:
        import h5py
    except ImportError:
        raise IOError('h5py is needed to open k-mer profile files and is not installed')
    return h5py.File(path, mode)


class FileType(object):
    """``argparse`` type for text files: ``-`` is stdin / stdout, existing files are never
    overwritten (kpal/__init__.py:47-80)."""

    def __init__(self, mode='r', bufsize=-1, encoding=None, errors=None):
        self._mode = mode
        self._bufsize = bufsize
        self._encoding = encoding
        self._errors = errors

    def __call__(self, string):
        if string == '-':
            if 'r' in self._mode:
                return sys.stdin
            if 'w' in self._mode:
                return sys.stdout
            raise ValueError('argument "-" with mode %r' % self._mode)
        try:
            if 'w' in self._mode and os.path.exists(string):
                raise OSError('file exists')
            return io.open(string, self._mode, self._bufsize, self._encoding, self._errors)
        except OSError as error:
            raise argparse.ArgumentTypeError("can't open '%s': %s" % (string, error))

    def __repr__(self):
        return '%s(%r)' % (type(self).__name__, self._mode)


class ProfileFileType(object):
    """``argparse`` type for k-mer profile files (kpal/__init__.py:83-111): a new file gets the format
    header and the ``profiles`` group, an existing one is refused for writing and checked for format
    and version for reading."""

    def __init__(self, mode='r'):
        self._mode = mode

    def __call__(self, string):
        try:
            if 'w' in self._mode and os.path.exists(string):
                raise IOError('file exists')
            handle = open_profile_file(string, self._mode)
            if 'w' in self._mode:
                handle.attrs['format'] = FORMAT_NAME
                handle.attrs['version'] = FORMAT_VERSION
                handle.attrs['producer'] = PRODUCER
                handle.create_group('profiles')
            else:
                fmt = handle.attrs.get('format')
                if isinstance(fmt, bytes):
                    fmt = fmt.decode('ascii', 'replace')
                if fmt != FORMAT_NAME:
                    raise IOError('not a k-mer profile file')
                version = handle.attrs['version']
                try:
                    accepted = format_version_accepted(version)
                except ValueError as error:
                    raise IOError(str(error))
                if not accepted:
                    raise IOError('file format version %s not supported' % (
                        version.decode() if isinstance(version, bytes) else version))
            return handle
        except IOError as error:
            raise argparse.ArgumentTypeError("can't open '%s': %s" % (string, error))

    def __repr__(self):
        return '%s(%s)' % (type(self).__name__, self._mode)


def doc_split(func):
    """First paragraph of a docstring: the sub-command descriptions."""
    return (func.__doc__ or '').split('\n\n')[0]


def version(name):
    return '{0} version {1} (command line and file format of kPAL 2.1; counting and distances on MI355X)'.format(
        name, __version__)
