"""A fixed-seed slice of the randomised parity stress tools (tests/stress_count.py, tests/stress_vec.py) in the driver's GPU
suite: random k, sizes from 1 B to 96 MB, compositions, separator densities, low-complexity stretches, host / device feeds in
one or several pieces, every strategy, count + balance -- and the vector side (balance, split, strand balance, summaries, every
ProfileDistance option combination, matrices) -- against the oracle.  The tools themselves run for minutes on other seeds when
a kernel changes; this is the slice that runs every time."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# (the long campaigns: tools/stress_campaign.sh, profiles/)
@pytest.mark.parametrize('tool,seconds,seed', [('stress_count.py', 15, 4), ('stress_vec.py', 10, 4)])
def test_stress_slice(tool, seconds, seed):
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', tool), '--seconds', str(seconds), '--seed', str(seed)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=480)
    out = p.stdout.decode()
    assert p.returncode == 0 and 'stress ok' in out, out[-3000:]
    cases = int(out.split('stress ok:')[1].split('cases')[0])
    assert cases >= 3, out[-500:]
