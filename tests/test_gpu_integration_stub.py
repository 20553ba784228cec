"""The reference-side binding of INTEGRATION.md, EXECUTED: the ```python block a kPAL maintainer would add as
``kpal/_hip.py`` is extracted from the document and run against libkpal_hip.so, then its ``count`` / ``balance`` /
``multiset`` are checked against the reference goldens G1 (kpal/klib.py:149-170), G5 (klib.py:285-298) and G6
(kpal/metrics.py:118-123) -- so a signature drift between include/kpal_hip.h and the documented stub fails a
test instead of a user.  Run on the GPU box: pytest -m gpu."""
import os
import re
import textwrap

import numpy as np
import pytest

from conftest import dense

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub_namespace(monkeypatch):
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, re.S)
    block = [b for b in blocks if 'ctypes.CDLL' in b]
    assert len(block) == 1, 'INTEGRATION.md must hold exactly one ctypes stub'
    monkeypatch.setenv('KPAL_HIP_LIBRARY', os.path.join(ROOT, 'kpal_amd', 'libkpal_hip.so'))
    ns = {}
    exec(compile(textwrap.dedent(block[0]), 'INTEGRATION.md', 'exec'), ns)
    return ns


def test_documented_stub_runs_and_matches_goldens(monkeypatch, golden_counts, golden_scalars, golden_vectors):
    hip = stub_namespace(monkeypatch)
    # every entry point the stub binds is declared in the header with the same arity
    header = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'kpal_hip.h')).read(), flags=re.S)
    for name in re.findall(r'_L\.(kpal_[a-z_]+)\.argtypes', open(os.path.join(ROOT, 'INTEGRATION.md')).read()):
        decl = re.search(r'\b%s\s*\(([^;]*?)\)\s*;' % name, header, re.S)
        assert decl, name
        assert len(getattr(hip['_L'], name).argtypes) == len([a for a in decl.group(1).split(',') if a.strip()]), name
    # G1: Profile.from_sequences on the reference's own fixtures
    for case in golden_counts['G1']:
        got = hip['count'](case['sequences'], case['k'])
        np.testing.assert_array_equal(got, dense(case['counts']))
    # G5: Profile.balance
    for rec in golden_scalars['G5']['balance_split']:
        v = golden_vectors['g5_%s_in' % rec['name']].copy()
        hip['balance'](v, rec['k'])
        np.testing.assert_array_equal(v, golden_vectors['g5_%s_bal' % rec['name']])
    # G6: metrics.multiset known answers (tests/test_metrics.py of the reference: 0.0625; left/right at k = 8)
    toy = golden_scalars['G6']['toy_k2']
    a = hip['count'](toy['a'], 2)
    b = hip['count'](toy['b'], 2)
    assert abs(hip['multiset'](a, b) - toy['distance']) <= 1e-9 * toy['distance']
    fx = {c['fixture']: c['sequences'] for c in golden_counts['G1']}
    left = hip['count'](fx['LENGTH_60'], 8)
    right = hip['count'](fx['LENGTH_60_MORE'], 8)
    want = golden_scalars['G6']['left_right_k8']
    assert abs(hip['multiset'](left, right) - want['prod']) <= 1e-9 * want['prod']            # tests/test_kdistlib.py:114-122
    assert abs(hip['multiset'](left, right, prod=False) - want['sum']) <= 1e-9 * want['sum']
    # bad k -> ValueError through the documented error mapping
    with pytest.raises(ValueError):
        hip['count'](['ACGT'], 17)
