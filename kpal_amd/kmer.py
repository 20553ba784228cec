"""Drop-in for ``kpal.kmer``: the seventeen command functions (kpal/kmer.py:52-700) and the command line
``main`` (kpal/kmer.py:703-975; ``python -m kpal_amd ...``).  They are orchestration: handles in, profiles
through :mod:`kpal_amd.klib` / :mod:`kpal_amd.kdistlib` (HIP kernels), text or an HDF5 handle out.  Same
sub-commands, arguments, defaults, output lines and ``ValueError`` messages as the reference.

Profile files are whatever the caller opens -- an ``h5py.File`` in kPAL; this module only uses the
handle operations kPAL itself uses (``handle['profiles']``, ``handle['profiles/<name>'][:]``,
``create_dataset``, ``attrs``, ``flush``).
"""
from __future__ import print_function

import importlib
import os
import re

import argparse
import sys

import numpy as np

from . import files, kdistlib, klib, metrics
from .files import FileType, ProfileFileType, doc_split

LENGTH_ERROR = 'k-mer lengths of the files differ'
NAMES_COUNT_ERROR = 'number of profile names does not match number of profiles'
PAIRED_NAMES_COUNT_ERROR = 'number of left and right profile names do not match'
PREFIX_COUNT_ERROR = 'number of name prefixes does not match number of profiles'

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


def convert(input_handles, output_handle, names=None):
    """Save k-mer profiles from files in the old plaintext format (kPAL < 1.0.0) to a k-mer profile file in
    the current HDF5 format.

    (kpal/kmer.py:52-74.)  Profiles are named by ``names``, else after the input files, else numbered."""
    names = names or [_name_from_handle(handle) for handle in input_handles]
    if len(names) != len(input_handles):
        raise ValueError(NAMES_COUNT_ERROR)
    for handle, name in zip(input_handles, names):
        klib.Profile.from_file_old_format(handle, name=name).save(output_handle)


def cat(input_handles, output_handle, names=None, prefixes=None):
    """Save k-mer profiles from several files to one k-mer profile file.

    (kpal/kmer.py:77-109.)  A name that a file does not hold is skipped for that file -- it may have been
    given to select from another one; ``prefixes`` (one per file) keep equal names apart."""
    prefixes = prefixes or ['' for _ in input_handles]
    if len(prefixes) != len(input_handles):
        raise ValueError(PREFIX_COUNT_ERROR)
    for handle, prefix in zip(input_handles, prefixes):
        for name in names or sorted(handle['profiles']):
            try:
                profile = klib.Profile.from_file(handle, name=name)
            except KeyError:
                continue
            profile.save(output_handle, name=prefix + name)


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


def distribution(input_handle, output_handle, names=None):
    """Calculate the distribution of the values in k-mer profiles: ``name count number-of-k-mers`` lines.

    (kpal/kmer.py:274-295.)"""
    for name in _profile_names(input_handle, names):
        profile = klib.Profile.from_file(input_handle, name=name)
        print('\n'.join('{0} {1} {2}'.format(name, value, number)
                        for value, number in metrics.distribution(profile.counts)), file=output_handle)


def info(input_handle, output_handle, names=None):
    """Print some information about k-mer profiles.

    (kpal/kmer.py:298-335.)  The five summaries of a profile come from one pass over it on the device."""
    names = _profile_names(input_handle, names)

    def text(value):
        return value.decode('utf-8', 'replace') if isinstance(value, bytes) else value

    print('File format version:', text(input_handle.attrs['version']), file=output_handle)
    print('Produced by:', text(input_handle.attrs['producer']), file=output_handle)
    for name in names:
        profile = klib.Profile.from_file(input_handle, name=name)
        stats = profile.summary()
        print('', file=output_handle)
        print('Profile:', profile.name, file=output_handle)
        print('- k-mer length:', str(profile.length), '({0} k-mers)'.format(profile.number), file=output_handle)
        print('- Zero counts:', str(profile.number - stats['non_zero']), file=output_handle)
        print('- Non-zero counts:', str(stats['non_zero']), file=output_handle)
        print('- Sum of counts:', str(stats['total']), file=output_handle)
        print('- Mean of counts:', '{0:.3f}'.format(stats['mean']), file=output_handle)
        print('- Median of counts:', '{0:.3f}'.format(stats['median']), file=output_handle)
        print('- Standard deviation of counts:', '{0:.3f}'.format(stats['std']), file=output_handle)


def get_count(input_handle, output_handle, word, names=None):
    """Retrieve the counts in k-mer profiles for a particular word.

    (kpal/kmer.py:338-362.)"""
    for name in _profile_names(input_handle, names):
        profile = klib.Profile.from_file(input_handle, name=name)
        if profile.length != len(word):
            raise ValueError('the length of the query does not match the profile length')
        try:
            offset = profile.dna_to_binary(word)
        except KeyError:
            raise ValueError('the input is not a valid DNA sequence')
        print(name, str(profile.counts[offset]), file=output_handle)


def _paired(input_handle_left, input_handle_right, names_left, names_right):
    """The (left, right) profile pairs of the two-file commands, linked by position in the name lists."""
    names_left = _profile_names(input_handle_left, names_left)
    names_right = _profile_names(input_handle_right, names_right)
    if len(names_left) != len(names_right):
        raise ValueError(PAIRED_NAMES_COUNT_ERROR)
    for name_left, name_right in zip(names_left, names_right):
        left = klib.Profile.from_file(input_handle_left, name=name_left)
        right = klib.Profile.from_file(input_handle_right, name=name_right)
        if left.length != right.length:
            raise ValueError(LENGTH_ERROR)
        yield left, right


def positive(input_handle_left, input_handle_right, output_handle_left, output_handle_right, names_left=None,
             names_right=None):
    """Only keep counts that are positive in both k-mer profiles; several profiles per file are linked by
    name and processed pairwise.

    (kpal/kmer.py:365-401.)"""
    for left, right in _paired(input_handle_left, input_handle_right, names_left, names_right):
        left.counts = metrics.positive(left.counts, right.counts)
        right.counts = metrics.positive(right.counts, left.counts)
        left.save(output_handle_left)
        right.save(output_handle_right)


def scale(input_handle_left, input_handle_right, output_handle_left, output_handle_right, names_left=None,
          names_right=None, down=False):
    """Scale two profiles such that the total number of k-mers is equal; several profiles per file are
    linked by name and processed pairwise.

    (kpal/kmer.py:404-444.)  The scaled counts are floats; ``save`` stores them as int64 like the reference."""
    for left, right in _paired(input_handle_left, input_handle_right, names_left, names_right):
        scale_left, scale_right = metrics.get_scale(left.counts, right.counts)
        if down:
            scale_left, scale_right = metrics.scale_down(scale_left, scale_right)
        left.counts = left.counts * scale_left
        right.counts = right.counts * scale_right
        left.save(output_handle_left)
        right.save(output_handle_right)


def shrink(input_handle, output_handle, factor, names=None):
    """Shrink k-mer profiles, effectively reducing k.

    (kpal/kmer.py:447-464.)"""
    for name in _profile_names(input_handle, names):
        profile = klib.Profile.from_file(input_handle, name=name)
        profile.shrink(factor)
        profile.save(output_handle)


def shuffle(input_handle, output_handle, names=None):
    """Randomise k-mer profiles.

    (kpal/kmer.py:467-483.)"""
    for name in _profile_names(input_handle, names):
        profile = klib.Profile.from_file(input_handle, name=name)
        profile.shuffle()
        profile.save(output_handle)


def smooth(input_handle_left, input_handle_right, output_handle_left, output_handle_right, names_left=None,
           names_right=None, summary='min', custom_summary=None, threshold=0):
    """Smooth two profiles by collapsing sub-profiles; several profiles per file are linked by name and
    processed pairwise.

    (kpal/kmer.py:486-538.)  Built-in summary functions run as the level-wise tree kernels."""
    names_left = _profile_names(input_handle_left, names_left)
    names_right = _profile_names(input_handle_right, names_right)
    if len(names_left) != len(names_right):      # before the custom function is evaluated, like the reference
        raise ValueError(PAIRED_NAMES_COUNT_ERROR)
    function = _custom_function(custom_summary, 'values') if custom_summary else metrics.summary[summary]
    dist = kdistlib.ProfileDistance(summary=function, threshold=threshold)
    for left, right in _paired(input_handle_left, input_handle_right, names_left, names_right):
        dist.dynamic_smooth(left, right)
        left.save(output_handle_left)
        right.save(output_handle_right)


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


def _parsers():
    """The option groups shared by the sub-commands (kpal/kmer.py:712-824), as argparse parents."""
    def parent():
        return argparse.ArgumentParser(add_help=False)

    p = {}
    p['multi_input'] = parent()
    p['multi_input'].add_argument('input_handles', metavar='INPUT', type=FileType('r'), nargs='*', default=[sys.stdin],
                                  help='input file (default: stdin)')
    p['input_profile'] = parent()
    p['input_profile'].add_argument('input_handle', metavar='INPUT', type=ProfileFileType('r'), help='input k-mer profile file')
    p['input_profile'].add_argument('-p', '--profiles', dest='names', metavar='NAME', type=str, nargs='+',
                                    help='names of the k-mer profiles to consider (default: all profiles in INPUT, in '
                                    'alphabetical order)')
    p['multi_input_profile'] = parent()
    p['multi_input_profile'].add_argument('input_handles', metavar='INPUT', type=ProfileFileType('r'), nargs='+',
                                          help='input k-mer profile file')
    p['multi_input_profile'].add_argument('-p', '--profiles', dest='names', metavar='NAME', type=str, nargs='+',
                                          help='names of the k-mer profiles to consider (default: all profiles per INPUT, '
                                          'in alphabetical order)')
    p['paired_input_profile'] = parent()
    p['paired_input_profile'].add_argument('input_handle_left', metavar='INPUT_LEFT', type=ProfileFileType('r'),
                                           help='input k-mer profile file (left)')
    p['paired_input_profile'].add_argument('input_handle_right', metavar='INPUT_RIGHT', type=ProfileFileType('r'),
                                           help='input k-mer profile file (right)')
    p['paired_input_profile'].add_argument('-l', '--profiles-left', dest='names_left', metavar='NAME', type=str, nargs='+',
                                           help='names of the k-mer profiles to consider (left) (default: all profiles '
                                           'in INPUT_LEFT, in alphabetical order)')
    p['paired_input_profile'].add_argument('-r', '--profiles-right', dest='names_right', metavar='NAME', type=str, nargs='+',
                                           help='names of the k-mer profiles to consider (right) (default: all profiles '
                                           'in INPUT_RIGHT, in alphabetical order)')
    p['output'] = parent()
    p['output'].add_argument('output_handle', metavar='OUTPUT', type=FileType('w'), help='output file')
    p['output_profile'] = parent()
    p['output_profile'].add_argument('output_handle', metavar='OUTPUT', type=ProfileFileType('w'),
                                     help='output k-mer profile file')
    p['paired_output_profile'] = parent()
    p['paired_output_profile'].add_argument('output_handle_left', metavar='OUTPUT_LEFT', type=ProfileFileType('w'),
                                            help='output k-mer profile file (left)')
    p['paired_output_profile'].add_argument('output_handle_right', metavar='OUTPUT_RIGHT', type=ProfileFileType('w'),
                                            help='output k-mer profile file (right)')
    p['scale'] = parent()
    p['scale'].add_argument('-d', dest='down', action='store_true', help='scale down')
    p['smooth'] = parent()
    p['smooth'].add_argument('-s', dest='summary', type=str, default='min', choices=metrics.summary,
                             help='summary function for dynamic smoothing (default: %(default)s)')
    p['smooth'].add_argument('-M', '--custom-summary', metavar='STRING', type=str, dest='custom_summary',
                             help='custom Python summary function, specified either by an expression over the NumPy '
                             'ndarray "values" (e.g., "np.max(values)"), or an importable name (e.g., '
                             '"package.module.summary") that can be called with an ndarray as argument')
    p['smooth'].add_argument('-t', dest='threshold', metavar='INT', type=int, default=0,
                             help='threshold for the summary function (default: %(default)s)')
    p['precision'] = parent()
    p['precision'].add_argument('-n', metavar='INT', dest='precision', type=int, default=10,
                                help='precision in number of decimals (default: %(default)s)')
    p['dist'] = argparse.ArgumentParser(add_help=False, parents=[p['scale'], p['smooth'], p['precision']])
    p['dist'].add_argument('-b', '--balance', dest='do_balance', action='store_true', help='balance the profiles')
    p['dist'].add_argument('--positive', dest='do_positive', action='store_true', help='use only positive values')
    p['dist'].add_argument('-S', '--scale', dest='do_scale', action='store_true', help='scale the profiles')
    p['dist'].add_argument('-m', '--smooth', dest='do_smooth', action='store_true', help='smooth the profiles')
    p['dist'].add_argument('-D', dest='distance_function', type=str, default='default', choices=metrics.vector_distance,
                           help='choose distance function (default: %(default)s)')
    p['dist'].add_argument('-P', dest='pairwise', type=str, default='prod', choices=metrics.pairwise,
                           help='paiwise distance function for the multiset distance (default: %(default)s)')
    p['dist'].add_argument('-f', '--pairwise-function', metavar='STRING', dest='custom_pairwise', type=str,
                           help='custom Python pairwise function, specified either by an expression over the two NumPy '
                           'ndarrays "left" and "right" (e.g., "abs(left - right) / (left + right + 1)"), or an importable '
                           'name (e.g., "package.module.pairwise") that can be called with two ndarrays as arguments')
    return p


def build_parser():
    """The ``kpal`` command line (kpal/kmer.py:826-964): seventeen sub-commands, same options and defaults."""
    p = _parsers()
    parser = argparse.ArgumentParser(formatter_class=argparse.RawDescriptionHelpFormatter, description=files.USAGE[0],
                                     epilog=files.USAGE[1])
    parser.add_argument('-v', action='version', version=files.version(parser.prog))
    subparsers = parser.add_subparsers(dest='subcommand')
    subparsers.required = True

    def command(name, func, parents, **defaults):
        sub = subparsers.add_parser(name, parents=[p[key] for key in parents], description=doc_split(func))
        sub.set_defaults(func=func, **defaults)
        return sub

    sub = command('convert', convert, ['multi_input', 'output_profile'])
    sub.add_argument('-p', '--profiles', dest='names', metavar='NAME', type=str, nargs='+',
                     help='names for the saved k-mer profiles, one per INPUT (default: profiles are named according to '
                     'the input filenames, or numbered consecutively from 1 if no filenames are available)')
    sub = command('cat', cat, ['multi_input_profile', 'output_profile'])
    sub.add_argument('-x', '--prefixes', dest='prefixes', metavar='PREFIX', type=str, nargs='+',
                     help='prefixes to use for the saved k-mer profile names, one per INPUT (default: profile names are '
                     'assumed to be disjoint and no prefix is used)')
    sub = command('count', count, ['multi_input', 'output_profile'])
    sub.add_argument('-p', '--profiles', dest='names', metavar='NAME', type=str, nargs='+',
                     help='names for the created k-mer profiles, one per INPUT (default: profiles are named according to '
                     'the input filenames, or numbered consecutively from 1 if no filenames are available)')
    sub.add_argument('-k', dest='size', metavar='SIZE', type=int, default=9, help='k-mer size (%(type)s default: %(default)s)')
    sub.add_argument('--by-record', '-r', dest='by_record', action='store_true',
                     help='make a k-mer profile per FASTA record instead of a k-mer profile per FASTA file (profiles are '
                     'named by the record names and prefixed according to --profiles if more than one INPUT is given)')
    sub = command('merge', merge, ['paired_input_profile', 'output_profile'])
    sub.add_argument('-m', dest='merger', type=str, default='sum', choices=metrics.mergers,
                     help='merge function (default: %(default)s)')
    sub.add_argument('-c', '--custom-merger', dest='custom_merger', metavar='STRING', type=str,
                     help='custom Python merge function, specified either by an expression over the two NumPy ndarrays '
                     '"left" and "right" (e.g., "np.add(left, right)"), or an importable name (e.g., '
                     '"package.module.merge") that can be called with two ndarrays as arguments')
    command('balance', balance, ['input_profile', 'output_profile'])
    command('showbalance', get_balance, ['input_profile', 'precision'], output_handle=sys.stdout)
    command('stats', get_stats, ['input_profile', 'precision'], output_handle=sys.stdout)
    command('distr', distribution, ['input_profile', 'output'])
    command('info', info, ['input_profile'], output_handle=sys.stdout)
    sub = command('getcount', get_count, ['input_profile'], output_handle=sys.stdout)
    sub.add_argument('word', metavar='WORD', type=str, help='the word in question')
    command('positive', positive, ['paired_input_profile', 'paired_output_profile'])
    command('scale', scale, ['paired_input_profile', 'paired_output_profile', 'scale'])
    sub = command('shrink', shrink, ['input_profile', 'output_profile'])
    sub.add_argument('-f', '--factor', dest='factor', metavar='INT', type=int, default=1,
                     help='shrinking factor (default: %(default)s)')
    command('shuffle', shuffle, ['input_profile', 'output_profile'])
    command('smooth', smooth, ['paired_input_profile', 'paired_output_profile', 'smooth'])
    command('distance', distance, ['paired_input_profile', 'dist'], output_handle=sys.stdout)
    command('matrix', distance_matrix, ['input_profile', 'output', 'dist'])
    return parser


def main(args=None):
    """Command line interface (kpal/kmer.py:703-975): ``args`` defaults to ``sys.argv[1:]``; a ``ValueError``
    of a command and an unreadable / existing file end in the parser's usage error (exit status 2)."""
    parser = build_parser()
    try:
        arguments = parser.parse_args(args)
    except IOError as error:
        parser.error(error)
    keywords = dict((key, value) for key, value in vars(arguments).items() if key not in ('func', 'subcommand'))
    try:
        arguments.func(**keywords)
    except ValueError as error:
        parser.error(error)
