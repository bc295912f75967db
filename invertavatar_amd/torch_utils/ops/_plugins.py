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


def _as_2d_filter(f, device):
    """None -> (NULL, 1, 1); [taps] -> outer product (the separable passes of upfirdn2d multiply out to it); [h, w] as is."""
    if f is None:
        return None, 1, 1
    f = f.to(device=device, dtype=torch.float32)
    if f.ndim == 1:
        f = f[:, None] * f[None, :]
    f = f.contiguous()
    return f, int(f.shape[0]), int(f.shape[1])


def filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filter, writeSigns):
    """filtered_lrelu_plugin.filtered_lrelu (torch_utils/ops/filtered_lrelu.cpp:20-22) -> (y, so, return_code).
    return_code -1 = "no kernel for this call" (sign tensors / float64 / oversized tile): the caller composes the op from
    bias_act + upfirdn2d exactly as the reference's Python does (filtered_lrelu.py:225-231)."""
    empty = torch.empty(0, device=x.device, dtype=torch.uint8)
    if (si is not None and si.numel()) or writeSigns or x.dtype not in (torch.float32, torch.float16) or x.ndim != 4:
        return None, empty, -1
    x = x.contiguous()
    fu2, fu_h, fu_w = _as_2d_filter(fu if (fu is None or fu.numel()) else None, x.device)
    fd2, fd_h, fd_w = _as_2d_filter(fd if (fd is None or fd.numel()) else None, x.device)
    n, c, ih, iw = x.shape
    oh = (ih * up + py0 + py1 - (fu_h - 1) - (fd_h - 1) + (down - 1)) // down
    ow = (iw * up + px0 + px1 - (fu_w - 1) - (fd_w - 1) + (down - 1)) // down
    if oh < 1 or ow < 1:
        raise RuntimeError('filtered_lrelu: output would be empty')
    if b is not None and b.numel():
        if b.dtype != x.dtype or b.numel() != c:
            raise RuntimeError('b must be a 1-D tensor of the dtype of x with one entry per channel')
        b = b.contiguous()
    else:
        b = None
    y = torch.empty(n, c, oh, ow, device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        st = _lib.load().ia_filtered_lrelu(x.data_ptr(), None if fu2 is None else fu2.data_ptr(), None if fd2 is None else fd2.data_ptr(),
                                           None if b is None else b.data_ptr(), y.data_ptr(), _lib.DTYPE_ID[x.dtype], n, c, ih, iw, oh, ow,
                                           fu_h, fu_w, fd_h, fd_w, int(up), int(down), int(px0), int(px1), int(py0), int(py1), float(gain),
                                           float(slope), float(clamp), 1 if flip_filter else 0, _lib.stream_ptr(x.device))
    if st == -2:          # IA_ERR_UNSUPPORTED: no kernel for this configuration
        return None, empty, -1
    _lib.check(st, 'filtered_lrelu')
    return y, empty, 0


def filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, writeSigns):
    """filtered_lrelu_plugin.filtered_lrelu_act_ (filtered_lrelu.cpp:217): the in-place activation with sign tensors that the
    reference's BACKWARD pass uses.  Gradients are outside this backend's scope (inference needs grad = 0)."""
    raise NotImplementedError('filtered_lrelu_act_ serves the gradient path only; differentiate through impl="ref"')


TABLE = {
    'bias_act_plugin': {'bias_act': bias_act},
    'upfirdn2d_plugin': {'upfirdn2d': upfirdn2d},
    'filtered_lrelu_plugin': {'filtered_lrelu': filtered_lrelu, 'filtered_lrelu_act_': filtered_lrelu_act_},
}
