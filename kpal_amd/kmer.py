"""Drop-in for the library functions of ``kpal.kmer`` that call the hot path:
``count``, ``merge``, ``balance``, ``get_balance``, ``get_stats``, ``distance`` and ``distance_matrix``
(kpal/kmer.py:112-271,541-700).  They are orchestration: handles in, profiles through
:mod:`kpal_amd.klib` / :mod:`kpal_amd.kdistlib` (HIP kernels), text or an HDF5 handle out.  Same
arguments, defaults, output lines and ``ValueError`` messages as the reference, so the reference's
``argparse`` front end (not rebuilt here) can call them unchanged.

Profile files are whatever the caller opens -- an ``h5py.File`` in kPAL; this module only uses the
handle operations kPAL itself uses (``handle['profiles']``, ``handle['profiles/<name>'][:]``,
``create_dataset``, ``attrs``, ``flush``).
"""
from __future__ import print_function

import importlib
import os
import re

import numpy as np

from . import kdistlib, klib, metrics

LENGTH_ERROR = 'k-mer lengths of the files differ'
NAMES_COUNT_ERROR = 'number of profile names does not match number of profiles'
PAIRED_NAMES_COUNT_ERROR = 'number of left and right profile names do not match'

# dotted path of an importable function, e.g. ``package.module.function`` (kpal/kmer.py:36-38)
_DOTTED_PATH = re.compile(r'[_a-zA-Z][_a-zA-Z0-9]*(\.[_a-zA-Z][_a-zA-Z0-9]*)+$')


def _name_from_handle(handle):
    """File name without directory and extension, or None for nameless handles and the
    ``<stdin>``-like ones (kpal/kmer.py:41-48)."""
    name = getattr(handle, 'name', None)
    if name is None or str(name).startswith('<'):
        return None
    return os.path.splitext(os.path.basename(str(name)))[0]


def _custom_function(definition, arguments):
    """A user-supplied function given on the command line: either a dotted import path or a Python
    expression over ``arguments`` with NumPy available as ``np`` (kpal/kmer.py:173-181,577-599).
    Such a callable never enters a kernel; klib / kdistlib run it through NumPy as the reference does."""
    if _DOTTED_PATH.match(definition):
        module, attribute = definition.rsplit('.', 1)
        return getattr(importlib.import_module(module), attribute)
    return eval('lambda %s: %s' % (arguments, definition), {'np': np})


def _profile_names(handle, names):
    return names or sorted(handle['profiles'])


def _fixed(precision, value):
    return '{{0:.{0}f}}'.format(precision).format(value)


def count(input_handles, output_handle, size, names=None, by_record=False):
    """k-mer profiles of FASTA files (kpal/kmer.py:112-146): one profile per file, or per record
    with ``by_record`` (record names, prefixed by the file's name when several files are given)."""
    names = names or [_name_from_handle(handle) for handle in input_handles]
    if len(names) != len(input_handles):
        raise ValueError(NAMES_COUNT_ERROR)
    several = len(input_handles) > 1
    for handle, name in zip(input_handles, names):
        if by_record:
            profiles = klib.Profile.from_fasta_by_record(handle, size, prefix=name if several else None)
        else:
            profiles = [klib.Profile.from_fasta(handle, size, name=name)]
        for profile in profiles:
            profile.save(output_handle)


def merge(input_handle_left, input_handle_right, output_handle, names_left=None, names_right=None, merger='sum',
          custom_merger=None):
    """Pairwise merge of the profiles of two files, linked by position in the (sorted) name lists
    (kpal/kmer.py:149-201); the result is named after both inputs."""
    names_left = _profile_names(input_handle_left, names_left)
    names_right = _profile_names(input_handle_right, names_right)
    if len(names_left) != len(names_right):
        raise ValueError(PAIRED_NAMES_COUNT_ERROR)
    function = _custom_function(custom_merger, 'left, right') if custom_merger else metrics.mergers[merger]
    for name_left, name_right in zip(names_left, names_right):
        left = klib.Profile.from_file(input_handle_left, name=name_left)
        right = klib.Profile.from_file(input_handle_right, name=name_right)
        if left.length != right.length:
            raise ValueError(LENGTH_ERROR)
        right.merge(left, function)      # merger(right, left), as the reference calls it
        right.save(output_handle, name=name_left if name_left == name_right else name_left + '_' + name_right)


def balance(input_handle, output_handle, names=None):
    """Balanced copies of the profiles of a file (kpal/kmer.py:203-219)."""
    for name in _profile_names(input_handle, names):
        profile = klib.Profile.from_file(input_handle, name=name)
        profile.balance()
        profile.save(output_handle)


def get_balance(input_handle, output_handle, precision=10, names=None):
    """``name balance`` lines: the multiset distance between the forward and the
    reverse-complement half of each profile (kpal/kmer.py:222-247) -- one fused kernel."""
    from . import _native
    for name in _profile_names(input_handle, names):
        profile = klib.Profile.from_file(input_handle, name=name)
        counts = np.asanyarray(profile.counts)
        if counts.dtype.kind in 'iub':
            score = _native.context().strand_balance(counts, profile.length, _native.PAIRWISE_PROD)
        else:
            forward, reverse = profile.split()
            score = metrics.multiset(forward, reverse, metrics.pairwise['prod'])
        print(name, _fixed(precision, score), file=output_handle)


def get_stats(input_handle, output_handle, precision=10, names=None):
    """``name mean std`` lines (kpal/kmer.py:250-271)."""
    for name in _profile_names(input_handle, names):
        profile = klib.Profile.from_file(input_handle, name=name)
        print(name, _fixed(precision, profile.mean), _fixed(precision, profile.std), file=output_handle)


def _profile_distance(distance_function, pairwise, custom_pairwise, do_smooth, summary, custom_summary, threshold,
                      do_scale, down, do_positive, do_balance):
    summary_function = _custom_function(custom_summary, 'values') if custom_summary else metrics.summary[summary]
    pairwise_function = (_custom_function(custom_pairwise, 'left, right') if custom_pairwise
                         else metrics.pairwise[pairwise])
    return kdistlib.ProfileDistance(do_balance=do_balance, do_positive=do_positive, do_smooth=do_smooth,
                                    summary=summary_function, threshold=threshold, do_scale=do_scale, down=down,
                                    pairwise=pairwise_function,
                                    distance_function=metrics.vector_distance[distance_function])


def distance(input_handle_left, input_handle_right, output_handle, names_left=None, names_right=None,
             distance_function='default', pairwise='prod', custom_pairwise=None, do_smooth=False, summary='min',
             custom_summary=None, threshold=0, do_scale=False, down=False, do_positive=False, do_balance=False,
             precision=10):
    """``left right distance`` lines for the profiles of two files, linked pairwise
    (kpal/kmer.py:541-620)."""
    names_left = _profile_names(input_handle_left, names_left)
    names_right = _profile_names(input_handle_right, names_right)
    if len(names_left) != len(names_right):
        raise ValueError(PAIRED_NAMES_COUNT_ERROR)
    dist = _profile_distance(distance_function, pairwise, custom_pairwise, do_smooth, summary, custom_summary,
                             threshold, do_scale, down, do_positive, do_balance)
    for name_left, name_right in zip(names_left, names_right):
        left = klib.Profile.from_file(input_handle_left, name=name_left)
        right = klib.Profile.from_file(input_handle_right, name=name_right)
        if left.length != right.length:
            raise ValueError(LENGTH_ERROR)
        print(name_left, name_right, _fixed(precision, dist.distance(left, right)), file=output_handle)


def distance_matrix(input_handle, output_handle, names=None, distance_function='default', pairwise='prod',
                    custom_pairwise=None, do_smooth=False, summary='min', custom_summary=None, threshold=0,
                    do_scale=False, down=False, do_positive=False, do_balance=False, precision=10):
    """Lower-triangular distance matrix of the profiles of a file (kpal/kmer.py:623-700); all pairs
    in one kernel launch for the built-in functions."""
    names = _profile_names(input_handle, names)
    if len(names) < 2:
        raise ValueError('you must give at least two k-mer profiles')
    dist = _profile_distance(distance_function, pairwise, custom_pairwise, do_smooth, summary, custom_summary,
                             threshold, do_scale, down, do_positive, do_balance)
    profiles = []
    for name in names:
        profiles.append(klib.Profile.from_file(input_handle, name=name))
        if profiles[0].length != profiles[-1].length:
            raise ValueError(LENGTH_ERROR)
    kdistlib.distance_matrix(profiles, output_handle, precision, dist)
