import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


class Golden:
    """Lazy view of one fixture file; arrays come back as torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name))

    def __getitem__(self, key):
        a = self._z[key]
        return torch.from_numpy(a) if a.ndim else a.item()

    def keys(self):
        return list(self._z.keys())


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def rnd(seed, *shape):
    """Same seeded inputs as tests/golden/make_golden.py:rnd."""
    return torch.from_numpy(np.random.RandomState(seed).randn(*shape).astype(np.float32))


def max_abs(a, b):
    return (a.double() - b.double()).abs().max().item()
