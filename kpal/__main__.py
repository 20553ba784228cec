"""``python -m kpal <sub-command> ...`` (the reference's console entry ``kpal = kpal.kmer:main``, setup.py:46-48)."""
from kpal_amd.kmer import main

if __name__ == '__main__':
    main()
