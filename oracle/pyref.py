"""Pure-Python restatement of the reference's counting loop, for ONE purpose: the like-for-like
1-core CPU figure of SURVEY.md 8d(1) (the reference itself never travels to the GPU box).

TEST INFRASTRUCTURE / BASELINE ONLY -- like everything under oracle/, nothing in kpal_amd imports it.
Follows kpal/klib.py:149-170 statement by statement: split every sequence at characters outside
AaCcGgTt, build the first k-mer of a part, then roll one character at a time; interpreter-speed by
design (that IS the reference's speed: 2.8 Mbases/s at k = 9 in the survey container).  Pinned by
tests/test_oracle_golden.py against golden G3 (BASELINE config 1) and the C oracle."""
import re

import numpy as np

_NUCLEOTIDE_TO_BINARY = {'A': 0, 'a': 0, 'C': 1, 'c': 1, 'G': 2, 'g': 2, 'T': 3, 't': 3}
_NOT_ALPHABET = re.compile('[^AaCcGgTt]')       # kpal/klib.py:152


def from_sequences(sequences, length):
    """int64[4**length] counts of all k-mers of an iterable of ``str`` (kpal/klib.py:149-170)."""
    number = 4 ** length
    bitmask = number - 1
    counts = [0] * number
    to_binary = _NUCLEOTIDE_TO_BINARY
    for sequence in sequences:
        for part in _NOT_ALPHABET.split(sequence):
            if len(part) >= length:
                binary = 0
                for c in part[:length]:
                    binary = (binary << 2) | to_binary[c]
                counts[binary] += 1
                for c in part[length:]:
                    binary = ((binary << 2) | to_binary[c]) & bitmask
                    counts[binary] += 1
    return np.array(counts, dtype='int64')
