"""Filtered leaky ReLU: bias -> up-FIR -> lrelu*gain -> clamp -> down-FIR (reference API:
torch_utils/ops/filtered_lrelu.py:58-118).

Not reached by the Next3D++ generator (only by the StyleGAN3 backbone).  Device fp32 / fp16 tensors run the fused
HIP kernel ``ia_filtered_lrelu`` (the up-sampled intermediate stays in LDS); everything else -- CPU tensors, float64, tensors
that need gradients -- is expressed through ``bias_act`` and ``upfirdn2d``, which is what the reference does when its fused
plugin reports "no kernel" (filtered_lrelu.py:225-231)."""
import numpy as np
import torch

from .. import misc
from . import _plugins, bias_act, upfirdn2d
from .upfirdn2d import _get_filter_size, _parse_padding


def _get_filter_size(f):  # noqa: F811  (same contract as upfirdn2d's; kept local for the API surface)
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and 1 <= f.ndim <= 2
    return int(f.shape[-1]), int(f.shape[0])


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='cuda'):
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if (impl == 'cuda' and x.device.type == 'cuda' and x.ndim == 4 and x.dtype in (torch.float32, torch.float16)
            and not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, fu, fd, b)))):
        px0, px1, py0, py1 = _parse_padding(padding)
        y, _, rc = _plugins.filtered_lrelu(x, fu, fd, b, None, int(up), int(down), px0, px1, py0, py1, 0, 0, float(gain), float(slope),
                                           float(-1 if clamp is None else clamp), bool(flip_filter), False)
        if rc == 0:
            return y
    return _filtered_lrelu_ref(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=padding, gain=gain, slope=slope,
                               clamp=clamp, flip_filter=flip_filter, impl=impl)


@misc.profiled_function
def _filtered_lrelu_ref(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                        flip_filter=False, impl='ref'):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    fu_w, fu_h = _get_filter_size(fu)
    fd_w, fd_h = _get_filter_size(fd)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.dtype == x.dtype
        misc.assert_shape(b, [x.shape[1]])
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    px0, px1, py0, py1 = _parse_padding(padding)
    assert gain == float(gain) and gain > 0
    assert slope == float(slope) and slope >= 0
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)
    n, c, in_h, in_w = x.shape
    out_w = (in_w * up + (px0 + px1) - (fu_w - 1) - (fd_w - 1) + (down - 1)) // down
    out_h = (in_h * up + (py0 + py1) - (fu_h - 1) - (fd_h - 1) + (down - 1)) // down
    dtype = x.dtype
    x = bias_act.bias_act(x=x, b=b, impl=impl)
    x = upfirdn2d.upfirdn2d(x=x, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter, impl=impl)
    x = bias_act.bias_act(x=x, act='lrelu', alpha=slope, gain=gain, clamp=clamp, impl=impl)
    x = upfirdn2d.upfirdn2d(x=x, f=fd, down=down, flip_filter=flip_filter, impl=impl)
    misc.assert_shape(x, [n, c, out_h, out_w])
    assert x.dtype == dtype
    return x
