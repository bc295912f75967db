"""Pad / up-sample / FIR-filter / down-sample (reference API: torch_utils/ops/upfirdn2d.py).

Public names and argument meaning follow the reference (setup_filter :72, upfirdn2d :120,
filter2d :279, upsample2d :315, downsample2d :354).  Device tensors run ``ia_upfirdn2d``."""
import dataclasses

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
        return _UpFirDn.apply(x, f, _FirPlan.parse(up, down, padding, flip_filter, gain))
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


@dataclasses.dataclass(frozen=True)
class _FirPlan:
    """One resampling step: zero-insert by `up`, pad by `pad` (x0, x1, y0, y1; negative crops), correlate with the (flipped unless
    `flip`) filter times `gain`, keep every `down`-th sample.  Hashable plain data: travels through autograd as a non-tensor argument."""
    up: tuple
    down: tuple
    pad: tuple
    flip: bool
    gain: float

    @staticmethod
    def parse(up, down, padding, flip_filter, gain):
        return _FirPlan(_parse_scaling(up), _parse_scaling(down), _parse_padding(padding), bool(flip_filter), gain)

    def adjoint(self, filter_wh, in_hw, out_hw):
        """Plan of the transposed operator (the input gradient): rates swapped, filter flipped the other way, and the padding that
        maps an `out_hw` gradient back onto `in_hw` samples."""
        (fw, fh), (ih, iw), (oh, ow) = filter_wh, in_hw, out_hw
        (ux, uy), (dx, dy), (px0, _, py0, _) = self.up, self.down, self.pad
        pad = (fw - px0 - 1, iw * ux - ow * dx + px0 - ux + 1, fh - py0 - 1, ih * uy - oh * dy + py0 - uy + 1)
        return _FirPlan(self.down, self.up, pad, not self.flip, self.gain)


def _launch(x, f, plan):
    """ia_upfirdn2d through the plugin-shaped entry; a 1-D filter is two launches (rows with unit gain, then columns)."""
    (ux, uy), (dx, dy), (px0, px1, py0, py1) = plan.up, plan.down, plan.pad
    if f.ndim == 2:
        return _plugin.upfirdn2d(x, f, ux, uy, dx, dy, px0, px1, py0, py1, plan.flip, plan.gain)
    y = _plugin.upfirdn2d(x, f.unsqueeze(0), ux, 1, dx, 1, px0, px1, 0, 0, plan.flip, 1.0)
    return _plugin.upfirdn2d(y, f.unsqueeze(1), 1, uy, 1, dy, 0, 0, py0, py1, plan.flip, plan.gain)


class _UpFirDn(torch.autograd.Function):
    """Device path of upfirdn2d().  The operator is linear in x, so its gradient is the same Function under the adjoint plan
    (and so on for higher orders); the filter gets no gradient (reference: upfirdn2d.py:250-273)."""

    @staticmethod
    def forward(ctx, x, f, plan):
        assert isinstance(x, torch.Tensor) and x.ndim == 4
        if f is None:
            f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
        elif f.ndim == 1 and f.shape[0] == 1:
            f = f.square().unsqueeze(0)          # one separable tap = one 2-D tap
        assert f.ndim in [1, 2]
        ctx.plan, ctx.in_hw = plan, tuple(x.shape[2:])
        ctx.save_for_backward(f)
        return _launch(x, f, plan)

    @staticmethod
    def backward(ctx, grad_out):
        assert not ctx.needs_input_grad[1], 'the FIR filter is a constant'
        if not ctx.needs_input_grad[0]:
            return None, None, None
        f, = ctx.saved_tensors
        back = ctx.plan.adjoint(_get_filter_size(f), ctx.in_hw, tuple(grad_out.shape[2:]))
        return _UpFirDn.apply(grad_out, f, back), None, None


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
