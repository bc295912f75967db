"""Pad / up-sample / FIR-filter / down-sample (reference API: torch_utils/ops/upfirdn2d.py).

Public names and argument meaning follow the reference (setup_filter :72, upfirdn2d :120,
filter2d :279, upsample2d :315, downsample2d :354).  Device tensors run ``ia_upfirdn2d``."""
import numpy as np
import torch

from .. import custom_ops, misc
from . import conv2d_gradfix

_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(module_name='upfirdn2d_plugin', sources=['upfirdn2d.hip'], headers=['ia_hip.h'])
    return True


def _parse_scaling(scaling):
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    assert isinstance(scaling, (list, tuple)) and all(isinstance(v, int) for v in scaling)
    sx, sy = scaling
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple)) and all(isinstance(v, int) for v in padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    assert fw >= 1 and fh >= 1
    return fw, fh


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """float32 FIR kernel: outer product for short 1-D taps, optional DC normalisation / flip / gain."""
    f = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    return (f * (gain ** (f.ndim / 2))).to(device=device)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _upfirdn2d_cuda(up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain).apply(x, f)
    return _upfirdn2d_ref(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)


@misc.profiled_function
def _upfirdn2d_ref(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """Plain-torch definition (CPU path of the reference, upfirdn2d.py:169-213)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32
    n, c, h, w = x.shape
    upx, upy = _parse_scaling(up)
    dnx, dny = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    assert w * upx + px0 + px1 >= f.shape[-1] and h * upy + py0 + py1 >= f.shape[0]
    z = x.new_zeros(n, c, h * upy, w * upx)
    z[:, :, ::upy, ::upx] = x
    z = torch.nn.functional.pad(z, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    z = z[:, :, max(-py0, 0):z.shape[2] - max(-py1, 0), max(-px0, 0):z.shape[3] - max(-px1, 0)]
    k = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:
        k = k.flip(list(range(k.ndim)))
    if k.ndim == 2:
        z = conv2d_gradfix.conv2d(z, k[None, None].repeat(c, 1, 1, 1), groups=c)
    else:
        z = conv2d_gradfix.conv2d(z, k[None, None, None, :].repeat(c, 1, 1, 1), groups=c)
        z = conv2d_gradfix.conv2d(z, k[None, None, :, None].repeat(c, 1, 1, 1), groups=c)
    return z[:, :, ::dny, ::dnx]


_cache = {}


def _upfirdn2d_cuda(up=1, down=1, padding=0, flip_filter=False, gain=1):
    upx, upy = _parse_scaling(up)
    dnx, dny = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    key = (upx, upy, dnx, dny, px0, px1, py0, py1, flip_filter, gain)
    if key in _cache:
        return _cache[key]

    class Upfirdn2dCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, f):
            assert isinstance(x, torch.Tensor) and x.ndim == 4
            if f is None:
                f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            if f.ndim == 1 and f.shape[0] == 1:
                f = f.square().unsqueeze(0)
            assert f.ndim in [1, 2]
            if f.ndim == 2:
                y = _plugin.upfirdn2d(x, f, upx, upy, dnx, dny, px0, px1, py0, py1, flip_filter, gain)
            else:  # separable: horizontal pass with unit gain, then vertical pass
                y = _plugin.upfirdn2d(x, f.unsqueeze(0), upx, 1, dnx, 1, px0, px1, 0, 0, flip_filter, 1.0)
                y = _plugin.upfirdn2d(y, f.unsqueeze(1), 1, upy, 1, dny, 0, 0, py0, py1, flip_filter, gain)
            ctx.save_for_backward(f)
            ctx.x_shape = x.shape
            return y

        @staticmethod
        def backward(ctx, dy):
            f, = ctx.saved_tensors
            _, _, ih, iw = ctx.x_shape
            _, _, oh, ow = dy.shape
            fw, fh = _get_filter_size(f)
            p = [fw - px0 - 1, iw * upx - ow * dnx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * dny + py0 - upy + 1]
            dx = None
            if ctx.needs_input_grad[0]:
                dx = _upfirdn2d_cuda(up=down, down=up, padding=p, flip_filter=(not flip_filter), gain=gain).apply(dy, f)
            assert not ctx.needs_input_grad[1]
            return dx, None

    _cache[key] = Upfirdn2dCuda
    return Upfirdn2dCuda


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    dnx, dny = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw - dnx + 1) // 2, px1 + (fw - dnx) // 2, py0 + (fh - dny + 1) // 2, py1 + (fh - dny) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
