"""kpal_amd -- MI355X-native k-mer counting and profile distance, API-compatible with the
``kpal.klib`` / ``kpal.metrics`` / ``kpal.kdistlib`` modules of LUMC/kPAL and the library functions of
``kpal.kmer`` that drive them.

    from kpal_amd import klib, metrics, kdistlib
    profile = klib.Profile.from_fasta(open('reads.fa'), 12)

The hot path lives in ``libkpal_hip.so`` (hand-written HIP for gfx950, C-ABI in
``include/kpal_hip.h``); build it with ``__graft_entry__.build()``.
"""
__version__ = '0.3.0'

from . import _native, metrics, klib, kdistlib, files, kmer, dist  # noqa: E402,F401
from .files import FileType, ProfileFileType  # noqa: F401
from .klib import Profile  # noqa: F401
from .kdistlib import ProfileDistance, distance_matrix  # noqa: F401
