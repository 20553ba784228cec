"""``python -m kpal_amd <sub-command> ...``: the kPAL command line (kpal/kmer.py:703-975, setup.py:46-48)."""
from .kmer import main

if __name__ == '__main__':
    main()
