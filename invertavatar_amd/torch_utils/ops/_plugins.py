"""Python-level plugin functions with the reference pybind signatures, implemented on
libia_hip.so (include/ia_hip.h).  They allocate the output (the reference plugins do that in
C++: bias_act.cpp:59, upfirdn2d.cpp:42) and pass raw device pointers + strides across the ABI."""
import torch

from ... import _lib


def _ptr(t):
    return t.data_ptr() if (t is not None and t.numel() > 0) else None


def _is_dense(t):
    """Non-overlapping and dense: some permutation of the dims is contiguous."""
    expect = 1
    for size, stride in sorted(((sz, st) for sz, st in zip(t.shape, t.stride()) if sz > 1), key=lambda p: p[1]):
        if stride != expect:
            return False
        expect *= size
    return True


def _check_device(t, what):
    if not t.is_cuda:
        raise RuntimeError(f'{what} must reside on the GPU (got {t.device})')


def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
    """bias_act_plugin.bias_act (bias_act.cpp:36)."""
    _check_device(x, 'x')
    if x.dtype not in _lib.DTYPE_ID:
        raise RuntimeError(f'bias_act: unsupported dtype {x.dtype}')
    if not _is_dense(x):
        raise RuntimeError('x must be non-overlapping and dense')
    has_b = b is not None and b.numel() > 0
    if has_b:
        if b.dtype != x.dtype or b.device != x.device:
            raise RuntimeError('b must have the same dtype and device as x')
        if b.ndim != 1:
            raise RuntimeError('b must have rank 1')
        if not (0 <= dim < x.ndim):
            raise RuntimeError('dim is out of bounds')
        if b.numel() != x.shape[dim]:
            raise RuntimeError('b has wrong number of elements')
        b = b.contiguous()
    for name, t in (('xref', xref), ('yref', yref), ('dy', dy)):
        if t is not None and t.numel() > 0 and (t.shape != x.shape or t.dtype != x.dtype or t.stride() != x.stride()):
            raise RuntimeError(f'{name} must have the same shape, dtype, and layout as x')
    y = torch.empty_like(x)  # preserves strides for dense tensors
    if y.stride() != x.stride():
        raise RuntimeError('y must have the same layout as x')
    if x.numel() == 0:
        return y
    lib = _lib.load()
    with torch.cuda.device(x.device):
        st = lib.ia_bias_act(_ptr(x), _ptr(b) if has_b else None, _ptr(xref), _ptr(yref), _ptr(dy), _ptr(y),
                             _lib.DTYPE_ID[x.dtype], x.numel(), b.numel() if has_b else 0,
                             x.stride(dim) if has_b else 1, int(grad), int(act), float(alpha), float(gain), float(clamp),
                             _lib.stream_ptr(x.device))
    _lib.check(st, 'bias_act')
    return y


def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
    """upfirdn2d_plugin.upfirdn2d (upfirdn2d.cpp:20)."""
    _check_device(x, 'x')
    if f.device != x.device:
        raise RuntimeError('f must reside on the same device as x')
    if f.dtype != torch.float32:
        raise RuntimeError('f must be float32')
    if x.ndim != 4:
        raise RuntimeError('x must be rank 4')
    if f.ndim != 2:
        raise RuntimeError('f must be rank 2')
    if x.numel() == 0:
        raise RuntimeError('x has zero size')
    if x.dtype not in _lib.DTYPE_ID:
        raise RuntimeError(f'upfirdn2d: unsupported dtype {x.dtype}')
    n, c, ih, iw = x.shape
    fh, fw = f.shape
    ow = (iw * upx + padx0 + padx1 - fw + downx) // downx
    oh = (ih * upy + pady0 + pady1 - fh + downy) // downy
    if ow < 1 or oh < 1:
        raise RuntimeError('output must be at least 1x1')
    fmt = torch.channels_last if (x.stride(1) == 1 and c > 1) else torch.contiguous_format
    y = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device, memory_format=fmt)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        st = lib.ia_upfirdn2d(_ptr(x), _ptr(f), _ptr(y), _lib.DTYPE_ID[x.dtype], n, c, ih, iw, _lib.strides64(x),
                              fh, fw, _lib.strides64(f), oh, ow, _lib.strides64(y),
                              int(upx), int(upy), int(downx), int(downy), int(padx0), int(pady0),
                              1 if flip else 0, float(gain), _lib.stream_ptr(x.device))
    _lib.check(st, 'upfirdn2d')
    return y


def _filtered_lrelu_unavailable(*args, **kwargs):
    raise NotImplementedError('filtered_lrelu has no HIP kernel yet (SURVEY.md 8f rank 2); use impl="ref"')


TABLE = {
    'bias_act_plugin': {'bias_act': bias_act},
    'upfirdn2d_plugin': {'upfirdn2d': upfirdn2d},
    'filtered_lrelu_plugin': {'filtered_lrelu': _filtered_lrelu_unavailable,
                              'filtered_lrelu_act_': _filtered_lrelu_unavailable},
}
