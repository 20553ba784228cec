import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _load(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


@pytest.fixture(scope='session')
def golden_counts():
    return _load('counts.json')


@pytest.fixture(scope='session')
def golden_synth():
    return _load('synth.json')['G3']


@pytest.fixture(scope='session')
def golden_scalars():
    return _load('scalars.json')


@pytest.fixture(scope='session')
def golden_vectors():
    return dict(np.load(os.path.join(GOLDEN, 'vectors.npz')))


def dense(sp):
    v = np.zeros(sp['n'], dtype=np.int64)
    v[np.asarray(sp['idx'], dtype=np.int64)] = np.asarray(sp['val'], dtype=np.int64)
    return v


@pytest.fixture(scope='session')
def tutorial_dir():
    return os.path.join(GOLDEN, 'tutorial')
