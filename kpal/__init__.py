"""``import kpal`` resolves to the MI355X implementation.

Import shim, nothing more: with this repository on ``sys.path`` ahead of LUMC/kPAL, code written against the reference --
``from kpal import klib``, ``from kpal.kdistlib import ProfileDistance``, the console entry ``kpal.kmer:main``
(setup.py:46-48 of the reference), ``python -m kpal count ...`` -- runs unmodified on the GPU: every sub-module name of the
reference package that lies on the hot path (kpal/klib.py, kpal/metrics.py, kpal/kdistlib.py, kpal/kmer.py) is an alias of
the module of the same name in :mod:`kpal_amd`, and the file types of ``kpal/__init__.py:47-111`` are re-exported.
``kpal.modality`` (outside the hot path) is not provided.
"""
import sys

import kpal_amd
from kpal_amd import kdistlib, klib, kmer, metrics                    # noqa: F401
from kpal_amd.files import FileType, ProfileFileType, doc_split, version   # noqa: F401

__version__ = kpal_amd.__version__

for _name in ('klib', 'metrics', 'kdistlib', 'kmer'):
    sys.modules[__name__ + '.' + _name] = getattr(kpal_amd, _name)
del _name
