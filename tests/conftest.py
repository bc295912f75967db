import contextlib
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


@contextlib.contextmanager
def fixed_randomness(jit, u_seed=99):
    """The two random draws of the renderer pinned exactly as tests/golden/make_golden.py:fixed_randomness pins them for the
    reference: torch.rand_like -> `jit` (stratified jitter), torch.rand(shape) -> RandomState(u_seed).rand(shape) (importance draws)."""
    orig_like, orig_rand = torch.rand_like, torch.rand

    def fake_like(t, *a, **k):
        assert tuple(t.shape) == tuple(jit.shape), (t.shape, jit.shape)
        return jit.to(device=t.device, dtype=t.dtype)

    def fake_rand(*size, **k):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        t = torch.from_numpy(np.random.RandomState(u_seed).rand(*shape).astype(np.float32))
        return t.to(k['device']) if k.get('device') is not None else t
    torch.rand_like, torch.rand = fake_like, fake_rand
    try:
        yield
    finally:
        torch.rand_like, torch.rand = orig_like, orig_rand
