"""Helpers used on the hot path (reference: torch_utils/misc.py:84 assert_shape,
:102 profiled_function, :157 copy_params_and_buffers).  Training-only helpers of the
reference file (samplers, DDP sync, module summary) are out of scope."""
import contextlib
import functools
import warnings

import torch


@contextlib.contextmanager
def suppress_tracer_warnings():
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', category=torch.jit.TracerWarning)
        yield


def assert_shape(tensor, ref_shape):
    """None entries of ref_shape are wildcards."""
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}')
    for idx, (size, ref) in enumerate(zip(tensor.shape, ref_shape)):
        if ref is not None and int(size) != int(ref):
            raise AssertionError(f'Wrong size for dimension {idx}: got {size}, expected {ref}')


def profiled_function(fn):
    """Wrap fn in a torch profiler range named after it (names kept so traces stay comparable)."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    return wrapper


def params_and_buffers(module):
    return list(module.parameters()) + list(module.buffers())


def named_params_and_buffers(module):
    return list(module.named_parameters()) + list(module.named_buffers())


@torch.no_grad()
def copy_params_and_buffers(src_module, dst_module, require_all=False, print_=True):
    """Copy tensors by name.  With require_all every destination tensor must exist in the
    source with a matching shape; otherwise mismatches are skipped (reference :157-184)."""
    src = dict(named_params_and_buffers(src_module))
    for name, dst in named_params_and_buffers(dst_module):
        if name not in src:
            if require_all:
                raise AssertionError(f'NotIn src_module {name}')
            continue
        if src[name].shape != dst.shape:
            if require_all:
                raise AssertionError(f'{name}: {tuple(src[name].shape)} vs {tuple(dst.shape)}')
            continue
        dst.copy_(src[name].detach()).requires_grad_(dst.requires_grad)
