"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU and exports
exactly the entry points include/ia_hip.h declares.  No compute calls here."""
import ctypes

import pytest
import torch

from invertavatar_amd import _lib, build


def test_library_builds_and_exports_header_symbols():
    path = build.build()
    lib = ctypes.CDLL(path)
    declared = _lib.declared_symbols()
    assert len(declared) >= 5
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/ia_hip.h but not exported'
    # every typed signature refers to a declared symbol
    assert set(_lib._SIGNATURES) <= set(declared), sorted(set(_lib._SIGNATURES) - set(declared))
    assert set(declared) <= set(_lib._SIGNATURES), sorted(set(declared) - set(_lib._SIGNATURES))


def test_version_and_error_text():
    lib = _lib.load()
    assert lib.ia_version() == 1
    # invalid argument is reported through the status code + ia_last_error, never an exception
    st = lib.ia_bias_act(None, None, None, None, None, None, 0, 16, 0, 1, 0, 3, 0.2, 1.0, -1.0, None)
    assert st == -1
    assert 'device pointers' in _lib.last_error()


def test_device_tensor_ops_never_fall_back(monkeypatch):
    """A CUDA tensor must go to the HIP kernel; a CPU tensor takes the reference's own 'ref' path."""
    from invertavatar_amd.torch_utils.ops import bias_act
    x = torch.randn(2, 3, 4, 4)
    y = bias_act.bias_act(x, torch.randn(3), act='lrelu')      # CPU tensor -> ref path, like the reference
    assert y.shape == x.shape
    with pytest.raises(RuntimeError):
        from invertavatar_amd.torch_utils.ops import _plugins
        _plugins.bias_act(x, torch.randn(3), None, None, None, 0, 1, 3, 0.2, 1.0, -1.0)  # CPU tensor at the plugin level
