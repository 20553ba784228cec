"""CPU parity oracle for kpal_amd (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; kpal_amd/ never does.  See kpal_oracle.c for the restated algorithm and its
reference citations.
"""
from .oracle import *  # noqa: F401,F403
