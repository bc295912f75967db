"""2-D convolution with optional FIR up/down-sampling (reference API:
torch_utils/ops/conv2d_resample.py:48-143).  Padding is applied once, with respect to the
up-sampled image.  Strategy per case:

  up > 1            stride-`up` transposed conv, then FIR with gain up**2 (and optional decimation)
  down > 1, 1x1     FIR+decimate first, then the 1x1 conv
  down > 1          FIR, then stride-`down` conv
  neither           plain conv when the padding is symmetric and non-negative
  otherwise         explicit up-FIR / conv / down-FIR
"""
import torch

from .. import misc
from . import conv2d_gradfix
from . import upfirdn2d
from .upfirdn2d import _get_filter_size, _parse_padding


def _get_weight_shape(w):
    return [int(s) for s in w.shape]


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """flip_weight=True is correlation (what F.conv2d does); False flips the taps first."""
    _, _, kh, kw = _get_weight_shape(w)
    if not flip_weight and (kw > 1 or kh > 1):
        w = w.flip([2, 3])
    if transpose:
        return conv2d_gradfix.conv_transpose2d(x, w, stride=stride, padding=padding, groups=groups)
    return conv2d_gradfix.conv2d(x, w, stride=stride, padding=padding, groups=groups)


@misc.profiled_function
def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    assert isinstance(groups, int) and groups >= 1
    out_ch, in_per_group, kh, kw = _get_weight_shape(w)
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)
    if up > 1:
        px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2; px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2; py1 += (fh - down) // 2
    pointwise = kw == 1 and kh == 1

    if pointwise and down > 1 and up == 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)

    if pointwise and up > 1 and down == 1:
        x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d.upfirdn2d(x=x, f=f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)

    if down > 1 and up == 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight)

    if up > 1:
        # transposed conv wants [in, out/groups, kh, kw]
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, out_ch // groups, in_per_group, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * in_per_group, out_ch // groups, kh, kw)
        px0 -= kw - 1; px1 -= kw - up
        py0 -= kh - 1; py1 -= kh - up
        pxt, pyt = max(min(-px0, -px1), 0), max(min(-py0, -py1), 0)
        x = _conv2d_wrapper(x=x, w=wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True,
                            flip_weight=(not flip_weight))
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2,
                                flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
        return x

    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:
        return _conv2d_wrapper(x=x, w=w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)

    x = upfirdn2d.upfirdn2d(x=x, f=(f if up > 1 else None), up=up, padding=[px0, px1, py0, py1], gain=up ** 2,
                            flip_filter=flip_filter)
    x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
    return x
