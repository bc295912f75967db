"""Oracle restatement of the operator layer.

Plain torch-CPU arithmetic, written from the formulas in SURVEY.md Appendix C and
checked against fixtures generated from the reference.  Test infrastructure only.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)

# activation name -> (default alpha, default gain); reference table: torch_utils/ops/bias_act.py:23-33
ACT_DEFAULTS = {
    'linear': (0.0, 1.0),
    'relu': (0.0, SQRT2),
    'lrelu': (0.2, SQRT2),
    'tanh': (0.0, 1.0),
    'sigmoid': (0.0, 1.0),
    'elu': (0.0, 1.0),
    'selu': (0.0, 1.0),
    'softplus': (0.0, 1.0),
    'swish': (0.0, SQRT2),
}


def _activate(x, act, alpha):
    if act == 'linear':
        return x
    if act == 'relu':
        return torch.clamp_min(x, 0)
    if act == 'lrelu':
        return torch.where(x >= 0, x, x * alpha)
    if act == 'tanh':
        return torch.tanh(x)
    if act == 'sigmoid':
        return torch.sigmoid(x)
    if act == 'elu':
        return F.elu(x)
    if act == 'selu':
        return F.selu(x)
    if act == 'softplus':
        return F.softplus(x)
    if act == 'swish':
        return torch.sigmoid(x) * x
    raise ValueError(act)


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """y = clamp(act(x + b) * gain).  Reference: torch_utils/ops/bias_act.py:93-122."""
    d_alpha, d_gain = ACT_DEFAULTS[act]
    alpha = d_alpha if alpha is None else float(alpha)
    gain = d_gain if gain is None else float(gain)
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    y = _activate(x, act, alpha)
    if gain != 1:
        y = y * gain
    if clamp is not None and clamp >= 0:
        y = y.clamp(-clamp, clamp)
    return y


def setup_filter(taps=(1, 3, 3, 1), normalize=True, flip_filter=False, gain=1.0, separable=None):
    """FIR set-up.  Reference: torch_utils/ops/upfirdn2d.py:72-116."""
    f = torch.as_tensor(taps, dtype=torch.float32)
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    return f * (gain ** (f.ndim / 2))


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _pad4(p):
    if isinstance(p, int):
        return p, p, p, p
    p = list(p)
    if len(p) == 2:
        return p[0], p[0], p[1], p[1]
    return tuple(p)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """Zero-insert, pad/crop, FIR, decimate.

    Reference: torch_utils/ops/upfirdn2d.py:169-213 (ref path) and the index maths of
    torch_utils/ops/upfirdn2d.cu:46-69.  Written as an explicit tap loop instead of the
    reference's depthwise conv2d so that it is an independent statement of the maths.
    """
    n, c, h, w = x.shape
    upx, upy = _pair(up)
    dnx, dny = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    if f is None:
        f = torch.ones(1, 1, dtype=torch.float32)
    if f.ndim == 1:  # separable: x pass with gain 1, then y pass with the gain
        y = upfirdn2d(x, f[None, :], (upx, 1), (dnx, 1), (px0, px1, 0, 0), flip_filter, 1.0)
        return upfirdn2d(y, f[:, None], (1, upy), (1, dny), (0, 0, py0, py1), flip_filter, gain)
    u = x.new_zeros(n, c, h * upy, w * upx)
    u[:, :, ::upy, ::upx] = x
    u = F.pad(u, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    u = u[:, :, max(-py0, 0):u.shape[2] - max(-py1, 0), max(-px0, 0):u.shape[3] - max(-px1, 0)]
    k = (f * gain).to(x.dtype)
    if not flip_filter:
        k = k.flip([0, 1])
    fh, fw = k.shape
    oh, ow = u.shape[2] - fh + 1, u.shape[3] - fw + 1
    acc = x.new_zeros(n, c, oh, ow)
    for ky in range(fh):
        for kx in range(fw):
            acc = acc + u[:, :, ky:ky + oh, kx:kx + ow] * k[ky, kx]
    return acc[:, :, ::dny, ::dnx]


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=2 ** 0.5, slope=0.2, clamp=None, flip_filter=False):
    """bias -> up-sample (up-FIR x up^2) -> lrelu(slope) x gain -> clamp -> down-FIR -> decimate.

    Reference: the op's own definition torch_utils/ops/filtered_lrelu.py:123-155 (_filtered_lrelu_ref), which is also the
    semantics of the fused CUDA kernel (filtered_lrelu.cu:144); composed here from the oracle's upfirdn2d / bias_act."""
    t = bias_act(x, b)
    t = upfirdn2d(t, fu, up=up, padding=padding, gain=float(up * up), flip_filter=flip_filter)
    t = bias_act(t, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    return upfirdn2d(t, fd, down=down, flip_filter=flip_filter)


def upsample2d(x, f, up=2, gain=1.0):
    """Reference: torch_utils/ops/upfirdn2d.py:315-350."""
    fh, fw = (f.shape[0], f.shape[-1]) if f.ndim == 2 else (f.shape[0], f.shape[0])
    p = [(fw + up - 1) // 2, (fw - up) // 2, (fh + up - 1) // 2, (fh - up) // 2]
    return upfirdn2d(x, f, up=up, padding=p, gain=gain * up * up)


def conv2d_up2(x, w, f):
    """3x3 conv with 2x up-sampling for ONE sample.

    x [1,I,H,W], w [O,I,3,3].  Reference: torch_utils/ops/conv2d_resample.py:114-131 with
    up=2, padding=1, flip_weight=False: stride-2 transposed conv (no spatial flip),
    then 4x4 FIR with pad [1,1,1,1] and gain up**2.
    """
    y = F.conv_transpose2d(x, w.transpose(0, 1), stride=2)
    return upfirdn2d(y, f, padding=(1, 1, 1, 1), gain=4.0)


def modulated_conv2d(x, weight, styles, noise=None, up=1, padding=0, resample_filter=None,
                     demodulate=True, fused=True):
    """Per-sample modulated conv.  Reference: training/networks_stylegan2.py:34-91.

    fused=True mirrors the grouped-conv branch (weights scaled per sample), fused=False
    the activation-scaling branch used when the generator is left in train() mode.
    Both are mathematically equal; the reference measures 3.6e-6 between them.
    """
    bsz = x.shape[0]
    o, i, kh, kw = weight.shape
    wmod = weight[None] * styles.reshape(bsz, 1, i, 1, 1)
    dcoef = None
    if demodulate:
        dcoef = (wmod.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()

    def conv_one(xb, wb):
        if up == 1:
            return F.conv2d(xb, wb, padding=padding)
        assert up == 2 and kh == 3 and padding == 1
        return conv2d_up2(xb, wb, resample_filter)

    outs = []
    for b in range(bsz):
        if fused:
            wb = wmod[b] * dcoef[b].reshape(o, 1, 1, 1) if demodulate else wmod[b]
            yb = conv_one(x[b:b + 1], wb)
        else:
            yb = conv_one(x[b:b + 1] * styles[b].reshape(1, i, 1, 1), weight)
            if demodulate:
                yb = yb * dcoef[b].reshape(1, o, 1, 1)
        outs.append(yb)
    y = torch.cat(outs, 0)
    if noise is not None:
        y = y + noise
    return y


def fully_connected(x, weight, bias, lr_multiplier=1.0, activation='linear'):
    """Reference: training/networks_stylegan2.py:96-127."""
    w = weight * (lr_multiplier / math.sqrt(weight.shape[1]))
    y = x @ w.t()
    b = bias * lr_multiplier if bias is not None else None
    if activation == 'linear':
        return y + b if b is not None else y
    return bias_act(y, b, act=activation)


def _aa_weights(n_in, n_out):
    """Dense [n_out, n_in] matrix of torch's anti-aliased bilinear resize along one
    axis (aten _upsample_bilinear2d_aa, align_corners=False)."""
    scale = n_in / n_out
    support = max(scale, 1.0)
    m = np.zeros((n_out, n_in), dtype=np.float64)
    for o in range(n_out):
        center = scale * (o + 0.5)
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), n_in)
        inv = 1.0 / support if scale >= 1.0 else 1.0
        ws = [max(0.0, 1.0 - abs((j - center + 0.5) * inv)) for j in range(lo, hi)]
        tot = sum(ws)
        for j, wgt in zip(range(lo, hi), ws):
            m[o, j] = wgt / tot
    return m


def resize_bilinear_aa(x, size):
    """F.interpolate(x, size, mode='bilinear', antialias=True) restated as two matmuls."""
    oh, ow = size
    _, _, h, w = x.shape
    if (oh, ow) == (h, w):
        return x.clone()
    mh = torch.from_numpy(_aa_weights(h, oh)).to(x.dtype)
    mw = torch.from_numpy(_aa_weights(w, ow)).to(x.dtype)
    y = torch.einsum('oh,nchw->ncow', mh, x)
    return torch.einsum('pw,ncow->ncop', mw, y)


def grid_sample_bilinear(img, grid):
    """F.grid_sample(img, grid, bilinear, zeros, align_corners=False).

    img [N,C,H,W], grid [N,Ho,Wo,2] (x to width).  Explicit 4-tap gather.
    """
    n, c, h, w = img.shape
    gx, gy = grid[..., 0], grid[..., 1]
    # aten's vectorised CPU kernel un-normalises as (g + 1) * (size / 2) - 0.5 (GridSamplerKernel.cpp,
    # ComputeLocation<align_corners=false>); algebraically equal to ((g + 1) * size - 1) / 2.
    px = (gx + 1) * (w / 2) - 0.5
    py = (gy + 1) * (h / 2) - 0.5
    x0 = torch.floor(px)
    y0 = torch.floor(py)
    fx = px - x0
    fy = py - y0
    x0 = x0.long()
    y0 = y0.long()
    out = img.new_zeros(n, c, *gx.shape[1:])
    flat = img.reshape(n, c, h * w)
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi, yi = x0 + dx, y0 + dy
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            idx = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).reshape(n, 1, -1).expand(-1, c, -1)
            tap = torch.gather(flat, 2, idx).reshape(n, c, *gx.shape[1:])
            out = out + tap * (wx * wy * ok.to(img.dtype)).unsqueeze(1)
    return out
