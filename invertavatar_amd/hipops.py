"""Tensor-level wrappers over the C ABI for the generator's fused HIP stages.

Each function allocates its output with torch (device memory is plumbing), passes raw
pointers to libia_hip.so on torch's current stream and raises on any non-zero status.
None of them has a CPU path: they are only ever called with device tensors."""
import contextlib
import contextvars
import ctypes
import math
import os

import torch

from . import _lib

ACT_ID = {'linear': 1, 'lrelu': 3}

# Optional per-launch timing for bench.py's roofline leg: when PROFILE is a list, every fused stage appends
# (kernel family, algorithmic FLOPs, algorithmic HBM bytes, start event, end event, shape note) recorded on the launch stream.
PROFILE = None


class _Timed:
    def __init__(self, family, flops, nbytes, desc=''):
        self.args = (family, float(flops), float(nbytes))
        self.desc = desc

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record()
            PROFILE.append(self.args + (self.e0, self.e1, self.desc))
        return False


def _p(t):
    return None if t is None else t.data_ptr()


def _f32c(t, what):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f'{what} must be a contiguous float32 device tensor')
    return t


def pack_conv_weight(w):
    """[O, I, kh, kw] -> [kh*kw, I, O] (tap-major, out-channel contiguous): the layout ia_conv2d_mfma reads."""
    o, i, kh, kw = w.shape
    return w.detach().permute(2, 3, 1, 0).reshape(kh * kw, i, o).contiguous()


def pack_conv_weight_h(w):
    """[O, I, kh, kw] -> fp16 [kh*kw, I/8, O, 8] (channel octets innermost): the layout ia_conv2d_mfma_h reads."""
    o, i, kh, kw = w.shape
    if i % 8:
        raise RuntimeError('fp16 packing needs in_channels % 8 == 0')
    return w.detach().permute(2, 3, 1, 0).reshape(kh * kw, i // 8, 8, o).permute(0, 1, 3, 2).contiguous().to(torch.float16)


def pack_conv_weight_split(w):
    """[O, I, kh, kw] -> fp16 [2, kh*kw, I/8, O, 8]: hi = fp16(w * 2^e) and lo = fp16(w * 2^e - hi), the layout ia_conv2d_mfma_s
    reads.  e (returned as the tensor attribute `wk_exp`) is the largest power of two with max|w| * 2^e <= 32768: it keeps the
    low parts, and the high parts times 2^-11, normal fp16 numbers (the MFMA flushes denormals)."""
    w = w.detach().float()
    top = float(w.abs().max())
    e = 0 if not (top > 0.0 and math.isfinite(top)) else max(-14, min(30, math.floor(math.log2(32768.0 / top))))
    ws = w * (2.0 ** e)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    out = torch.stack([pack_conv_weight_h(hi), pack_conv_weight_h(lo)]).contiguous()
    out.wk_exp = int(e)
    return out


def conv_h_supported(i, o, h, w, ksize, transposed):
    """Shapes ia_conv2d_mfma_h covers (see include/ia_hip.h)."""
    if ksize != 3 or i % 8 or o % 4:
        return False
    if transposed:
        return (h + 1) * (w + 1) > 320
    return o >= 128 and h * w >= 1024 and w <= 512


def conv_sx_supported(i, o, h, w, ksize, transposed):
    """Shapes the split-DMA form (ia_conv2d_mfma_sx) covers: the library's own rule (ia_conv2d_sx_supported) -- 3x3 layers from 8^2 up
    (the 4^2 layers stay on the fp32 tiles: no gain measured)."""
    return bool(_lib.load().ia_conv2d_sx_supported(int(i), int(o), int(h), int(w), int(ksize), int(bool(transposed))))


def weight_sq_sum(w):
    """wsq[o, i] = sum over taps of w^2: the static half of the demodulation reduction."""
    return w.detach().float().square().sum(dim=[2, 3]).contiguous()


def modconv_demod(styles, wsq):
    b, i = styles.shape
    o = wsq.shape[0]
    d = torch.empty(b, o, device=styles.device, dtype=torch.float32)
    st = _lib.load().ia_modconv_demod(_p(_f32c(styles, 'styles')), _p(_f32c(wsq, 'wsq')), _p(d), b, i, o, _lib.stream_ptr(styles.device))
    _lib.check(st, 'ia_modconv_demod')
    return d


# Stream-K accumulator slabs (caller-owned memory, as the ABI requires).  Launches on different streams may overlap, so a slab
# is keyed by (device, stream).  A captured graph may be replayed on any stream, concurrently with eager launches or with other
# graphs, so each GraphedSynthesis brings its OWN registry (scratch_owner) for the duration of its capture; the registry dies
# with the graph object.  The default registry serves eager launches.
_default_scratch = {}
_scratch_registry = contextvars.ContextVar('ia_scratch_registry', default=None)


@contextlib.contextmanager
def scratch_owner(registry):
    """Route the scratch allocations of the launches issued inside the block to `registry` (a dict the caller keeps alive)."""
    token = _scratch_registry.set(registry)
    try:
        yield registry
    finally:
        _scratch_registry.reset(token)


def _scratch_buffer(device, nbytes):
    """Grow-only scratch for the stream-K accumulator slabs, one per (device, stream) of the active registry."""
    reg = _scratch_registry.get()
    if reg is None:
        reg = _default_scratch
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = reg.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((nbytes + 3) // 4, device=device, dtype=torch.float32)
        reg[key] = buf
    return buf


def conv2d_mfma(x, wk, styles=None, demod=None, noise=None, noise_strength=None, bias=None, residual=None,
                ksize=3, transposed=False, act='linear', alpha=0.2, gain=1.0, clamp=None, ksplit=None):
    """One fused StyleGAN2 convolution (see ia_conv2d_mfma in include/ia_hip.h).  A float16 `wk` (pack_conv_weight_h)
    selects the fp16-operand form ia_conv2d_mfma_h."""
    _f32c(x, 'x')
    b, i, h, w = x.shape
    half_ops = wk.dtype == torch.float16
    split = half_ops and wk.dim() == 5
    if half_ops:
        if not (wk.is_cuda and wk.is_contiguous() and wk.dim() in (4, 5) and wk.shape[-1] == 8 and (not split or wk.shape[0] == 2)):
            raise RuntimeError('wk must be a contiguous fp16 [taps, I/8, O, 8] (or [2, taps, I/8, O, 8] hi/lo) device tensor')
        if split and not hasattr(wk, 'wk_exp'):
            raise RuntimeError('hi/lo weights must come from pack_conv_weight_split (they carry their scale as .wk_exp)')
        taps, wi, o = wk.shape[-4], wk.shape[-3] * 8, wk.shape[-2]
    else:
        _f32c(wk, 'wk')
        taps, wi, o = wk.shape
    if taps != ksize * ksize or wi != i:
        raise RuntimeError(f'packed weight {tuple(wk.shape)} does not match ksize {ksize}, in-channels {i}')
    for name, t in (('styles', styles), ('demod', demod), ('noise', noise), ('bias', bias), ('residual', residual)):
        if t is not None:
            _f32c(t, name)
    lib = _lib.load()
    plan_s, plan_bytes = ctypes.c_int(0), ctypes.c_size_t(0)
    form = 2 if split else (1 if half_ops else 0)
    _lib.check(lib.ia_conv2d_plan(b, i, o, h, w, ksize, int(transposed), form, ctypes.byref(plan_s), ctypes.byref(plan_bytes)), 'ia_conv2d_plan')
    nbytes = plan_bytes.value
    if ksplit is None:
        ksplit = plan_s.value
    elif plan_s.value > 0 and ksplit != plan_s.value:   # caller-chosen worker count (tests): scale the slabs with it
        nbytes = nbytes // plan_s.value * max(int(ksplit), 1)
    oh, ow = (2 * h + 1, 2 * w + 1) if transposed else (h, w)
    y = torch.empty(b, o, oh, ow, device=x.device, dtype=torch.float32)
    scratch = None
    if nbytes:
        scratch = _scratch_buffer(x.device, nbytes)
    flops = 2.0 * b * h * w * i * o * ksize * ksize
    traffic = 4.0 * (x.numel() + wk.numel() + y.numel() + (residual.numel() if residual is not None else 0))
    with torch.cuda.device(x.device), _Timed('conv2d_mfma_t' if transposed else f'conv2d_mfma_k{ksize}', flops, traffic,
                                             f'B{b} I{i} O{o} {h}x{w} G{ksplit}' + (' f16x3' if split else (' f16' if half_ops else ''))):
        fn = lib.ia_conv2d_mfma_s if split else (lib.ia_conv2d_mfma_h if half_ops else lib.ia_conv2d_mfma)
        head = (_p(x), _p(wk), int(wk.wk_exp)) if split else (_p(x), _p(wk))
        st = fn(*head, _p(styles), _p(demod), _p(noise), _p(noise_strength), _p(bias), _p(residual),
                                _p(y), _p(scratch), nbytes, b, i, o, h, w, ksize, int(transposed), ACT_ID[act], float(alpha),
                                float(gain), float(-1 if clamp is None else clamp), int(ksplit), _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_conv2d_mfma')
    return y


def conv1x1_supported(i, o, h, w):
    """Shapes ia_conv1x1 covers (see include/ia_hip.h)."""
    return o <= 96 and i % 32 == 0 and (h * w) % 4 == 0


def conv1x1(x, wk, styles=None, bias=None, residual=None, clamp=None):
    """ToRGB layer in one streaming launch (see ia_conv1x1): clamp((wk * styles) (*) x + bias) + residual.  wk [1, I, O] or [I, O]
    (pack_conv_weight of a 1x1 weight)."""
    _f32c(x, 'x')
    _f32c(wk, 'wk')
    b, i, h, w = x.shape
    o = wk.shape[-1]
    if wk.numel() != i * o:
        raise RuntimeError(f'wk has {wk.numel()} elements, expected {i} x {o}')
    for t, name, n in ((styles, 'styles', b * i), (bias, 'bias', o), (residual, 'residual', b * o * h * w)):
        if t is not None and (_f32c(t, name).numel() != n):
            raise RuntimeError(f'{name} has {t.numel()} elements, expected {n}')
    y = torch.empty(b, o, h, w, device=x.device, dtype=torch.float32)
    flops = 2.0 * b * h * w * i * o
    traffic = 4.0 * (x.numel() + wk.numel() + y.numel() + (residual.numel() if residual is not None else 0))
    with torch.cuda.device(x.device), _Timed('conv1x1', flops, traffic, f'B{b} I{i} O{o} {h}x{w}'):
        st = _lib.load().ia_conv1x1(_p(x), _p(wk), _p(styles), _p(bias), _p(residual), _p(y), b, i, o, h, w,
                                    float(-1 if clamp is None else clamp), _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_conv1x1')
    return y


def torgb_supported(i, o, h, w, with_skip=False):
    """Shapes ia_torgb covers (see include/ia_hip.h)."""
    return o <= 96 and i in (128, 256, 512, 1024) and not (with_skip and (h % 2 or w % 2))


def torgb(x, wk, styles=None, bias=None, residual=None, skip=None, skip_filter=None, clamp=None):
    """ToRGB layer + the skip-image branch in one launch (see ia_torgb): clamp((wk * styles) (*) x + bias) + residual, or
    + upsample2d(skip, skip_filter) with skip [B, O, H/2, W/2] and the [4, 4] resample filter."""
    _f32c(x, 'x')
    _f32c(wk, 'wk')
    b, i, h, w = x.shape
    o = wk.shape[-1]
    if wk.numel() != i * o:
        raise RuntimeError(f'wk has {wk.numel()} elements, expected {i} x {o}')
    if residual is not None and skip is not None:
        raise RuntimeError('either a residual image or a skip image, not both')
    for t, name, n in ((styles, 'styles', b * i), (bias, 'bias', o), (residual, 'residual', b * o * h * w),
                       (skip, 'skip', b * o * (h // 2) * (w // 2)), (skip_filter, 'skip_filter', 16)):
        if t is not None and (_f32c(t, name).numel() != n):
            raise RuntimeError(f'{name} has {t.numel()} elements, expected {n}')
    if skip is not None and skip_filter is None:
        raise RuntimeError('skip needs skip_filter (the [4, 4] resample filter)')
    y = torch.empty(b, o, h, w, device=x.device, dtype=torch.float32)
    flops = 2.0 * b * h * w * i * o
    traffic = 4.0 * (x.numel() + wk.numel() + y.numel() + (residual.numel() if residual is not None else 0) + (skip.numel() if skip is not None else 0))
    with torch.cuda.device(x.device), _Timed('conv1x1', flops, traffic, f'B{b} I{i} O{o} {h}x{w} torgb'):
        st = _lib.load().ia_torgb(_p(x), _p(wk), _p(styles), _p(bias), _p(residual), _p(skip), _p(skip_filter), _p(y), b, i, o, h, w,
                                  float(-1 if clamp is None else clamp), _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_torgb')
    return y


# Debug switch: every producer of a SplitAct counts the elements its hi / lo split clamped at +-65504 (ia_split_saturation_count) and
# raises when there are any -- the range contract of the fp16-pair convolutions, otherwise silent (one sync per producer: tests only).
CHECK_SPLIT_RANGE = False


def split_saturation_poll(device=None, reset=True):
    """True when any hi / lo split on `device` since the last reset met a value outside +-65504 (it was clamped): the always-on
    range watch of the fp16-pair convolutions (ia_split_saturation_poll; a device -> host read, not for captured regions)."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    flag = ctypes.c_uint(0)
    with torch.cuda.device(device):
        st = _lib.load().ia_split_saturation_poll(ctypes.byref(flag), int(bool(reset)), _lib.stream_ptr(device))
    _lib.check(st, 'ia_split_saturation_poll')
    return flag.value != 0


def split_saturation_clear(device=None):
    """Clear the range-watch flags of `device` in stream order; no host synchronisation (the top of a clip / an inversion)."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(device):
        st = _lib.load().ia_split_saturation_poll(None, 1, _lib.stream_ptr(device))
    _lib.check(st, 'ia_split_saturation_poll')


def split_saturation_count(sa):
    """Elements of the SplitAct's hi plane that sit on the fp16 maximum (a device -> host sync)."""
    b, c, h, w = sa.shape
    cnt = torch.zeros(1, device=sa.data.device, dtype=torch.int32)
    with torch.cuda.device(sa.data.device):
        st = _lib.load().ia_split_saturation_count(_p(sa.data), int(sa.planes), b, c, h, w, _p(cnt), _lib.stream_ptr(sa.data.device))
    _lib.check(st, 'ia_split_saturation_count')
    return int(cnt.item())


class SplitAct:
    """An activation stored as fp16 planes [B, planes, C/8, H, W, 8] (the input format of ia_conv2d_mfma_sx, see include/ia_hip.h):
    planes = 2 hi / lo pairs (fp32-equivalent consumers), planes = 1 one rounded fp16 plane (fp16-operand consumers: the
    fp16-storage form); already multiplied by the styles of the layer `consumer` (any object; None = unscaled)."""

    requires_grad = False      # (forward-only: lets the autograd eligibility checks treat it like a detached tensor)

    def __init__(self, data, channels, consumer=None):
        self.data, self.channels, self.consumer = data, channels, consumer
        b, self.planes, c8, h, w, _ = data.shape
        self.shape = (b, channels, h, w)
        self.device, self.dtype = data.device, torch.float32      # stands in for an fp32 activation
        if CHECK_SPLIT_RANGE and not torch.cuda.is_current_stream_capturing():
            n = split_saturation_count(self)
            if n:
                raise OverflowError(f'{n} activations of a {self.shape} split tensor were clamped at +-65504 (fp16 range of the hi plane)')

    def float(self):     # inverse of the split (tests / fallbacks): hi + lo * 2^-11, channel groups unfolded
        v = self.data[:, 0].float()
        if self.planes == 2:
            v = v + self.data[:, 1].float() * (1.0 / 2048.0)
        b, c8, h, w, _ = v.shape
        return v.permute(0, 1, 4, 2, 3).reshape(b, c8 * 8, h, w)


def act_split(x, styles=None, consumer=None, planes=2, shift=None):
    """fp32 NCHW (x styles [B,C] + shift [B,C]) -> SplitAct (see ia_act_split)."""
    _f32c(x, 'x')
    b, c, h, w = x.shape
    if c % 8:
        raise RuntimeError('the split format needs channels % 8 == 0')
    out = torch.empty(b, planes, c // 8, h, w, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device), _Timed('act_split', 0.0, (4.0 + 2.0 * planes) * x.numel()):
        st = _lib.load().ia_act_split(_p(x), _p(None if styles is None else _f32c(styles, 'styles')),
                                      _p(None if shift is None else _f32c(shift, 'shift')), _p(out), int(planes), b, c, h, w,
                                      _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_act_split')
    return SplitAct(out, c, consumer)


def bn_train_split(x, weight, bias, running_mean, running_var, num_batches_tracked, eps, momentum, consumer=None, planes=2):
    """ia_bn_train_split: train-mode BatchNorm2d (batch statistics over B, H, W; running statistics moved in place when given) folded
    into the split staging of the convolution that follows it.  Returns the SplitAct of the normalised tensor."""
    _f32c(x, 'x')
    b, c, h, w = x.shape
    if c % 8:
        raise RuntimeError(f'the split format stores channels in groups of 8 (C = {c})')
    for name, t in (('weight', weight), ('bias', bias), ('running_mean', running_mean), ('running_var', running_var)):
        if t is not None and _f32c(t, name).numel() != c:
            raise RuntimeError(f'{name} has {t.numel()} elements, expected {c}')
    if num_batches_tracked is not None and not (num_batches_tracked.is_cuda and num_batches_tracked.dtype == torch.int64 and num_batches_tracked.numel() == 1):
        raise RuntimeError('num_batches_tracked must be a device int64 scalar')
    chunks = max(1, min(256, -(-(b * h * w) // 8192)))
    partials = torch.empty(c * chunks * 2, device=x.device, dtype=torch.float64)
    out = torch.empty(b, planes, c // 8, h, w, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device), _Timed('bn_train_split', 3.0 * x.numel(), 4.0 * x.numel() * 2 + 2.0 * out.numel(), f'B{b} C{c} {h}x{w}'):
        st = _lib.load().ia_bn_train_split(_p(x), _p(weight), _p(bias), _p(running_mean), _p(running_var), _p(num_batches_tracked), _p(partials), chunks,
                                           _p(out), int(planes), b, c, h, w, float(eps), float(momentum), _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_bn_train_split')
    for t in (running_mean, running_var, num_batches_tracked):      # written through raw pointers: tell torch (the eval-mode affine caches of
        if t is not None:                                           # trunk_hip._UnitPack are keyed on the version counters)
            torch.autograd.graph.increment_version(t)
    return SplitAct(out, c, consumer)


def conv2d_mfma_sx(xs, wk, demod=None, noise=None, noise_strength=None, bias=None, residual=None, transposed=False, act='linear',
                   alpha=0.2, gain=1.0, clamp=None, want_f32=True, split_for=None, styles_next=None, ksplit=None, split_planes=2, prelu=None,
                   want_split=None):
    """ia_conv2d_mfma_sx: 3x3 convolution of a SplitAct (already multiplied by this layer's styles).  Returns the fp32 result,
    a SplitAct for `split_for` (multiplied by styles_next, `split_planes` planes), or the pair (y, ys) when both are asked for.
    A two-plane input takes weights from pack_conv_weight_split, a one-plane input those of pack_conv_weight_h."""
    if not isinstance(xs, SplitAct):
        raise RuntimeError('xs must be a SplitAct (hipops.act_split or a producing layer)')
    if xs.planes == 2 and not (wk.dtype == torch.float16 and wk.dim() == 5 and wk.shape[0] == 2 and hasattr(wk, 'wk_exp')):
        raise RuntimeError('wk must come from pack_conv_weight_split')
    if xs.planes == 1 and not (wk.dtype == torch.float16 and wk.dim() == 4):
        raise RuntimeError('a one-plane input takes the weights of pack_conv_weight_h')
    b, i, h, w = xs.shape
    o = wk.shape[-2]
    if wk.shape[-4] != 9 or wk.shape[-3] * 8 != i:
        raise RuntimeError(f'packed weight {tuple(wk.shape)} does not match 3x3, in-channels {i}')
    if want_split is None:
        want_split = split_for is not None or styles_next is not None
    if prelu is not None and (_f32c(prelu, 'prelu').numel() != o or act != 'lrelu'):
        raise RuntimeError('prelu: [O] per-channel slopes, with act="lrelu"')
    if transposed and (want_split or not want_f32):
        raise RuntimeError('the transposed form only writes the fp32 image')
    for name, t in (('demod', demod), ('noise', noise), ('bias', bias), ('residual', residual), ('styles_next', styles_next)):
        if t is not None:
            _f32c(t, name)
    lib = _lib.load()
    plan_s, plan_bytes = ctypes.c_int(0), ctypes.c_size_t(0)
    _lib.check(lib.ia_conv2d_plan(b, i, o, h, w, 3, int(transposed), 3, ctypes.byref(plan_s), ctypes.byref(plan_bytes)), 'ia_conv2d_plan')
    nbytes = plan_bytes.value
    if ksplit is None:
        ksplit = plan_s.value
    elif plan_s.value > 0 and ksplit != plan_s.value:
        nbytes = nbytes // plan_s.value * max(int(ksplit), 1)
    oh, ow = (2 * h + 1, 2 * w + 1) if transposed else (h, w)
    dev = xs.data.device
    y = torch.empty(b, o, oh, ow, device=dev, dtype=torch.float32) if want_f32 else None
    ys = torch.empty(b, split_planes, o // 8, oh, ow, 8, device=dev, dtype=torch.float16) if want_split else None
    scratch = _scratch_buffer(dev, nbytes) if nbytes else None
    flops = 2.0 * b * h * w * i * o * 9
    traffic = (2.0 * (xs.data.numel() + wk.numel()) + 4.0 * (y.numel() if want_f32 else 0) + 2.0 * (ys.numel() if want_split else 0)
               + 4.0 * (residual.numel() if residual is not None else 0))
    with torch.cuda.device(dev), _Timed('conv2d_mfma_t' if transposed else 'conv2d_mfma_k3', flops, traffic,
                                        f'B{b} I{i} O{o} {h}x{w} G{ksplit} ' + ('f16x3 dma' if xs.planes == 2 else 'f16 dma')):
        st = lib.ia_conv2d_mfma_sx(_p(xs.data), int(xs.planes), _p(wk), int(getattr(wk, 'wk_exp', 0)), _p(demod), _p(noise), _p(noise_strength),
                                   _p(bias), _p(residual), _p(y), _p(ys), int(split_planes), _p(styles_next), _p(scratch), nbytes, b, i, o, h, w, int(transposed), ACT_ID[act],
                                   float(alpha), _p(prelu), float(gain), float(-1 if clamp is None else clamp), int(ksplit), _lib.stream_ptr(dev))
    _lib.check(st, 'ia_conv2d_mfma_sx')
    out_s = SplitAct(ys, o, split_for) if want_split else None
    if want_f32 and want_split:
        return y, out_s
    return y if want_f32 else out_s


def upsample_bilinear_add(x, addend):
    """ia_upsample_bilinear_add: bilinear resize (align_corners=True) of x [B, C, H, W] to addend's size, plus addend."""
    _f32c(x, 'x')
    _f32c(addend, 'addend')
    b, c, h, w = x.shape
    if addend.shape[:2] != x.shape[:2]:
        raise RuntimeError(f'upsample_bilinear_add: {tuple(x.shape)} onto {tuple(addend.shape)}')
    y = torch.empty_like(addend)
    with torch.cuda.device(x.device):
        st = _lib.load().ia_upsample_bilinear_add(_p(x), _p(addend), _p(y), b * c, h, w, addend.shape[2], addend.shape[3], _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_upsample_bilinear_add')
    return y


def dwconv3x3_tokens(x, w9c, bias, h, w, gelu=False):
    """ia_dwconv3x3_tokens: depth-wise 3x3 convolution of tokens [B, H*W, C] on their H x W grid (+ GELU); w9c [9, C]."""
    _f32c(x, 'x')
    _f32c(w9c, 'w9c')
    b, n, c = x.shape
    if n != h * w or tuple(w9c.shape) != (9, c) or c % 4:
        raise RuntimeError(f'dwconv3x3_tokens: tokens {tuple(x.shape)} on a {h} x {w} grid with weights {tuple(w9c.shape)} (C % 4 == 0)')
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = _lib.load().ia_dwconv3x3_tokens(_p(x), _p(w9c), _p(None if bias is None else _f32c(bias, 'bias')), _p(y), b, h, w, c, 1 if gelu else 0,
                                             _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_dwconv3x3_tokens')
    return y


def dwconv3x3_tokens_split(x, w9c, bias, h, w, gelu=False):
    """dwconv3x3_tokens with the result written as SplitTokens (the operand of the linear layer behind it; see ia_dwconv3x3_tokens_split)."""
    _f32c(x, 'x')
    _f32c(w9c, 'w9c')
    b, n, c = x.shape
    if n != h * w or tuple(w9c.shape) != (9, c) or c % 16:
        raise RuntimeError(f'dwconv3x3_tokens_split: tokens {tuple(x.shape)} on a {h} x {w} grid with weights {tuple(w9c.shape)} (C % 16 == 0)')
    out = torch.empty(2, c // 8, b * n, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device):
        st = _lib.load().ia_dwconv3x3_tokens_split(_p(x), _p(w9c), _p(None if bias is None else _f32c(bias, 'bias')), _p(out), b, h, w, c,
                                                   1 if gelu else 0, _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_dwconv3x3_tokens_split')
    return SplitTokens(out, b * n, c, (b, n))


def conv_tiny_supported(i, o, h, w):
    """Shapes ia_conv3x3_s2_tiny covers (3x3, stride 2, padding 1 on 2^2 / 4^2 / 8^2 images)."""
    return bool(_lib.load().ia_conv3x3_s2_tiny_supported(int(i), int(o), int(h), int(w)))


def conv3x3_s2_tiny(x, w, bias=None, act='linear', alpha=0.2):
    """ia_conv3x3_s2_tiny: [B, I, H, W] (H = W in {2, 4, 8}) -> [B, O, H/2, W/2] with the module's own weight [O, I, 3, 3]."""
    _f32c(x, 'x')
    _f32c(w, 'w')
    b, i, h, wd = x.shape
    o = w.shape[0]
    if tuple(w.shape) != (o, i, 3, 3):
        raise RuntimeError(f'weight {tuple(w.shape)} does not match [O, {i}, 3, 3]')
    if bias is not None and _f32c(bias, 'bias').numel() != o:
        raise RuntimeError(f'bias has {bias.numel()} elements, expected {o}')
    y = torch.empty(b, o, h // 2, wd // 2, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device), _Timed('conv3x3_tiny', 2.0 * b * (h // 2) * (wd // 2) * i * o * 9, 4.0 * (x.numel() + w.numel() + y.numel()),
                                             f'B{b} I{i} O{o} {h}x{wd} s2'):
        st = _lib.load().ia_conv3x3_s2_tiny(_p(x), _p(w), _p(bias), _p(y), b, i, o, h, wd, ACT_ID[act], float(alpha), _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_conv3x3_s2_tiny')
    return y


def conv_down_supported(b, i, o, h, w):
    """Shapes ia_conv2d_down_sx covers (3x3, stride 2, padding 1 on an h x w input): asks the library's planner."""
    ks, nbytes = ctypes.c_int(0), ctypes.c_size_t(0)
    return _lib.load().ia_conv2d_down_plan(int(b), int(i), int(o), int(h), int(w), ctypes.byref(ks), ctypes.byref(nbytes)) == 0


def conv2d_down_sx(xs, wk, demod=None, bias=None, residual=None, act='linear', alpha=0.2, gain=1.0, clamp=None, want_f32=True, split_for=None,
                   styles_next=None, split_planes=2, prelu=None, want_split=None):
    """ia_conv2d_down_sx: 3x3 convolution with stride 2 and padding 1 of a SplitAct -> [B, O, (H-1)//2+1, (W-1)//2+1]; results as
    conv2d_mfma_sx returns them."""
    if not isinstance(xs, SplitAct):
        raise RuntimeError('xs must be a SplitAct (hipops.act_split or a producing layer)')
    if xs.planes == 2 and not (wk.dtype == torch.float16 and wk.dim() == 5 and wk.shape[0] == 2 and hasattr(wk, 'wk_exp')):
        raise RuntimeError('wk must come from pack_conv_weight_split')
    if xs.planes == 1 and not (wk.dtype == torch.float16 and wk.dim() == 4):
        raise RuntimeError('a one-plane input takes the weights of pack_conv_weight_h')
    b, i, h, w = xs.shape
    o = wk.shape[-2]
    if wk.shape[-4] != 9 or wk.shape[-3] * 8 != i:
        raise RuntimeError(f'packed weight {tuple(wk.shape)} does not match 3x3, in-channels {i}')
    if want_split is None:
        want_split = split_for is not None or styles_next is not None
    if prelu is not None and (_f32c(prelu, 'prelu').numel() != o or act != 'lrelu'):
        raise RuntimeError('prelu: [O] per-channel slopes, with act="lrelu"')
    for name, t in (('demod', demod), ('bias', bias), ('residual', residual), ('styles_next', styles_next)):
        if t is not None:
            _f32c(t, name)
    lib = _lib.load()
    plan_s, plan_bytes = ctypes.c_int(0), ctypes.c_size_t(0)
    _lib.check(lib.ia_conv2d_down_plan(b, i, o, h, w, ctypes.byref(plan_s), ctypes.byref(plan_bytes)), 'ia_conv2d_down_plan')
    nbytes, ksplit = plan_bytes.value, plan_s.value
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    dev = xs.data.device
    y = torch.empty(b, o, oh, ow, device=dev, dtype=torch.float32) if want_f32 else None
    ys = torch.empty(b, split_planes, o // 8, oh, ow, 8, device=dev, dtype=torch.float16) if want_split else None
    scratch = _scratch_buffer(dev, nbytes) if nbytes else None
    flops = 2.0 * b * oh * ow * i * o * 9
    traffic = (2.0 * (xs.data.numel() + wk.numel()) + 4.0 * (y.numel() if want_f32 else 0) + 2.0 * (ys.numel() if want_split else 0)
               + 4.0 * (residual.numel() if residual is not None else 0))
    with torch.cuda.device(dev), _Timed('conv2d_mfma_k3', flops, traffic, f'B{b} I{i} O{o} {h}x{w} s2 G{ksplit} ' + ('f16x3 dma' if xs.planes == 2 else 'f16 dma')):
        st = lib.ia_conv2d_down_sx(_p(xs.data), int(xs.planes), _p(wk), int(getattr(wk, 'wk_exp', 0)), _p(demod), _p(bias), _p(residual), _p(y), _p(ys),
                                   int(split_planes), _p(styles_next), _p(scratch), nbytes, b, i, o, h, w, ACT_ID[act], float(alpha), _p(prelu),
                                   float(gain), float(-1 if clamp is None else clamp), int(ksplit), _lib.stream_ptr(dev))
    _lib.check(st, 'ia_conv2d_down_sx')
    out_s = SplitAct(ys, o, split_for) if want_split else None
    if want_f32 and want_split:
        return y, out_s
    return y if want_f32 else out_s


def upconv_rows_supported(b, i, o, h, w):
    """Shapes ia_upconv2d_rows_sx covers (the row-phase form of the transposed 3x3 convolution): asks the library."""
    nbytes = ctypes.c_size_t(0)
    return _lib.load().ia_upconv2d_rows_plan(int(b), int(i), int(o), int(h), int(w), ctypes.byref(nbytes)) == 0


def upconv2d_rows_sx(xs, wk, demod=None):
    """ia_upconv2d_rows_sx: stride-2 transposed 3x3 convolution of a two-plane SplitAct -> the demodulated [B, O, 2H+1, 2W+1] fp32 image
    (what conv2d_mfma_sx(transposed=True) returns), evaluated per output row phase on the stride-1 tile."""
    if not (isinstance(xs, SplitAct) and xs.planes == 2):
        raise RuntimeError('xs must be a two-plane SplitAct')
    if not (wk.dtype == torch.float16 and wk.dim() == 5 and wk.shape[0] == 2 and hasattr(wk, 'wk_exp')):
        raise RuntimeError('wk must come from pack_conv_weight_split')
    b, i, h, w = xs.shape
    o = wk.shape[-2]
    if wk.shape[-4] != 9 or wk.shape[-3] * 8 != i:
        raise RuntimeError(f'packed weight {tuple(wk.shape)} does not match 3x3, in-channels {i}')
    if demod is not None:
        _f32c(demod, 'demod')
    lib = _lib.load()
    plan_bytes = ctypes.c_size_t(0)
    _lib.check(lib.ia_upconv2d_rows_plan(b, i, o, h, w, ctypes.byref(plan_bytes)), 'ia_upconv2d_rows_plan')
    dev = xs.data.device
    y = torch.empty(b, o, 2 * h + 1, 2 * w + 1, device=dev, dtype=torch.float32)
    scratch = _scratch_buffer(dev, plan_bytes.value)
    flops = 2.0 * b * h * w * i * o * 9
    traffic = 2.0 * (xs.data.numel() + wk.numel()) + 4.0 * y.numel()
    with torch.cuda.device(dev), _Timed('conv2d_mfma_t', flops, traffic, f'B{b} I{i} O{o} {h}x{w} rows f16x3 dma'):
        st = lib.ia_upconv2d_rows_sx(_p(xs.data), _p(wk), int(wk.wk_exp), _p(demod), _p(y), _p(scratch), plan_bytes.value, b, i, o, h, w,
                                     _lib.stream_ptr(dev))
    _lib.check(st, 'ia_upconv2d_rows_sx')
    return y


def conv_sx_rgb_supported(b, i, o, h, w):
    """Layers ia_conv2d_mfma_sx_rgb covers: stride-1 layers that run in whole rounds of tiles holding every output channel."""
    if o > 128 or i % 8 or o % 8:
        return False
    plan_s, plan_bytes = ctypes.c_int(0), ctypes.c_size_t(0)
    st = _lib.load().ia_conv2d_plan(b, i, o, h, w, 3, 0, 3, ctypes.byref(plan_s), ctypes.byref(plan_bytes))
    # (whole rounds of the 128-channel x 256-point tile: a layer with fewer of those than CUs runs on the 32-channel tile family or stream-K)
    return st == 0 and plan_s.value == 0 and h * w >= 8192 and b * ((h * w + 255) // 256) >= 256


def conv2d_mfma_sx_rgb(xs, wk, rgb_wk, rgb_styles=None, rgb_bias=None, rgb_residual=None, rgb_clamp=None, demod=None, noise=None,
                       noise_strength=None, bias=None, act='linear', alpha=0.2, gain=1.0, clamp=None, want_f32=False, split_for=None,
                       styles_next=None, split_planes=2):
    """ia_conv2d_mfma_sx_rgb: the stride-1 3x3 convolution of a SplitAct AND the ToRGB layer that reads its result, in one launch.
    rgb_wk = pack_conv_weight of the 1x1 ToRGB weight ([1, O, RC] or [O, RC]).  Returns (y or None, SplitAct or None, rgb)."""
    if not isinstance(xs, SplitAct):
        raise RuntimeError('xs must be a SplitAct (hipops.act_split or a producing layer)')
    b, i, h, w = xs.shape
    o = wk.shape[-2]
    if wk.shape[-4] != 9 or wk.shape[-3] * 8 != i:
        raise RuntimeError(f'packed weight {tuple(wk.shape)} does not match 3x3, in-channels {i}')
    _f32c(rgb_wk, 'rgb_wk')
    rc = rgb_wk.shape[-1]
    if rgb_wk.numel() != o * rc:
        raise RuntimeError(f'rgb_wk has {rgb_wk.numel()} elements, expected {o} x {rc}')
    for name, t, n in (('demod', demod, b * o), ('noise', noise, h * w), ('bias', bias, o), ('styles_next', styles_next, b * o),
                       ('rgb_styles', rgb_styles, b * o), ('rgb_bias', rgb_bias, rc), ('rgb_residual', rgb_residual, b * rc * h * w)):
        if t is not None and _f32c(t, name).numel() != n:
            raise RuntimeError(f'{name} has {t.numel()} elements, expected {n}')
    want_split = split_for is not None or styles_next is not None
    dev = xs.data.device
    y = torch.empty(b, o, h, w, device=dev, dtype=torch.float32) if want_f32 else None
    ys = torch.empty(b, split_planes, o // 8, h, w, 8, device=dev, dtype=torch.float16) if want_split else None
    rgb = torch.empty(b, rc, h, w, device=dev, dtype=torch.float32)
    flops = 2.0 * b * h * w * i * o * 9
    traffic = (2.0 * (xs.data.numel() + wk.numel()) + 4.0 * (y.numel() if want_f32 else 0) + 2.0 * (ys.numel() if want_split else 0)
               + 4.0 * rgb.numel() * (2 if rgb_residual is not None else 1))
    with torch.cuda.device(dev), _Timed('conv2d_mfma_k3', flops, traffic, f'B{b} I{i} O{o} {h}x{w} G0 ' + ('f16x3 dma' if xs.planes == 2 else 'f16 dma') + ' +rgb'):
        st = _lib.load().ia_conv2d_mfma_sx_rgb(_p(xs.data), int(xs.planes), _p(wk), int(getattr(wk, 'wk_exp', 0)), _p(demod), _p(noise),
                                               _p(noise_strength), _p(bias), _p(y), _p(ys), int(split_planes), _p(styles_next), _p(rgb_wk),
                                               _p(rgb_styles), _p(rgb_bias), _p(rgb_residual), _p(rgb), int(rc),
                                               float(-1 if rgb_clamp is None else rgb_clamp), b, i, o, h, w, ACT_ID[act], float(alpha), float(gain),
                                               float(-1 if clamp is None else clamp), _lib.stream_ptr(dev))
    _lib.check(st, 'ia_conv2d_mfma_sx_rgb')
    return y, (SplitAct(ys, o, split_for) if want_split else None), rgb


def upfirdn2d_bias_act(x, f, noise=None, noise_strength=None, bias=None, up=1, pad0=(1, 1), out_hw=None, fir_gain=1.0,
                       act='linear', alpha=0.2, act_gain=1.0, clamp=None, flip=False):
    """FIR + noise + bias + activation in one pass (see ia_upfirdn2d_bias_act)."""
    if not (x.is_cuda and x.is_contiguous() and x.dtype in (torch.float32, torch.float16)):
        raise RuntimeError('x must be a contiguous f32/f16 device tensor')
    n, c, ih, iw = x.shape
    fh, fw = f.shape
    y = torch.empty(n, c, out_hw[0], out_hw[1], device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device), _Timed('upfirdn2d_bias_act', 2.0 * y.numel() * (fh * fw) / (up * up),
                                             x.element_size() * (x.numel() + y.numel())):
        st = _lib.load().ia_upfirdn2d_bias_act(_p(x), _p(_f32c(f, 'f')), _p(noise), _p(noise_strength), _p(bias), _p(y),
                                               _lib.DTYPE_ID[x.dtype], n, c, ih, iw, fh, fw, out_hw[0], out_hw[1], int(up),
                                               int(pad0[0]), int(pad0[1]), 1 if flip else 0, float(fir_gain), ACT_ID[act],
                                               float(alpha), float(act_gain), float(-1 if clamp is None else clamp),
                                               _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_upfirdn2d_bias_act')
    return y


def fir_tail_split(x, f, noise=None, noise_strength=None, bias=None, styles_next=None, out_hw=None, pad0=(1, 1), fir_gain=1.0, act='linear',
                   alpha=0.2, act_gain=1.0, clamp=None, flip=False, want_f32=False, split_for=None, planes=2):
    """upfirdn2d_bias_act (4x4 filter, up 1) whose result goes out in split format for `split_for` (see ia_fir_tail_split).
    Returns the SplitAct, or (y, SplitAct) with want_f32."""
    _f32c(x, 'x')
    n, c, ih, iw = x.shape
    if tuple(f.shape) != (4, 4) or c % 8:
        raise RuntimeError('fir_tail_split needs the 4x4 filter and channels % 8 == 0')
    oh, ow = out_hw
    ys = torch.empty(n, planes, c // 8, oh, ow, 8, device=x.device, dtype=torch.float16)
    y = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.float32) if want_f32 else None
    with torch.cuda.device(x.device), _Timed('upfirdn2d_bias_act', 2.0 * n * c * oh * ow * 16, 4.0 * (x.numel() + n * c * oh * ow * (2 if want_f32 else 1)),
                                             'split'):
        st = _lib.load().ia_fir_tail_split(_p(x), _p(_f32c(f, 'f')), _p(noise), _p(noise_strength), _p(bias), _p(styles_next), _p(y), _p(ys),
                                           int(planes), n, c, ih, iw, oh, ow, int(pad0[0]), int(pad0[1]), 1 if flip else 0, float(fir_gain), ACT_ID[act],
                                           float(alpha), float(act_gain), float(-1 if clamp is None else clamp), _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_fir_tail_split')
    out = SplitAct(ys, c, split_for)
    return (y, out) if want_f32 else out


def compose_upfir_weight(w, f, gain=4.0):
    """[O, I, 3, 3] weight of an up-sampling layer + its [4, 4] resample filter -> [4*O, I, 3, 3]: per output phase (py, px) the 3x3
    kernel on the INPUT image that equals conv_transpose2d(stride 2) followed by upfirdn2d(f, padding [1,1,1,1], gain)
    (see ia_upconv2d_fir_sx).  Row ((py * O/32 + o // 32) * 2 + px) * 32 + o % 32."""
    w = w.detach().double()
    f = torch.as_tensor(f).double().to(w.device)
    if tuple(f.shape) != (4, 4) or tuple(w.shape[2:]) != (3, 3):
        raise RuntimeError('compose_upfir_weight: 3x3 weight and 4x4 filter')
    F = f.flip([0, 1])                                   # upfirdn2d applies the flipped filter (a true convolution)
    o, i = w.shape[:2]
    out = torch.zeros(4, o, i, 3, 3, dtype=torch.float64, device=w.device)
    for py in range(2):
        for px in range(2):
            for a in range(4):
                for ky in range(3):
                    ny = py + a - 1 - ky
                    if ny % 2:
                        continue
                    for b_ in range(4):
                        for kx in range(3):
                            nx = px + b_ - 1 - kx
                            if nx % 2:
                                continue
                            out[2 * py + px, :, :, ny // 2 + 1, nx // 2 + 1] += gain * F[a, b_] * w[:, :, ky, kx]
    if o % 32:
        raise RuntimeError('compose_upfir_weight: out_channels % 32 == 0')
    # row order (py, block of 32 channels, px, channel in block): the two horizontal phases of a block are neighbours (see the kernel's store)
    out = out.view(2, 2, o // 32, 32, i, 3, 3).permute(0, 2, 1, 3, 4, 5, 6)
    return out.reshape(4 * o, i, 3, 3).float()


def upconv_fir_supported(b, i, o, h, w):
    """Layers ia_upconv2d_fir_sx covers: the composed 4*O-channel stride-1 layer runs in whole tiles that divide O."""
    if i % 8 or o % 128:
        return False
    plan_s, plan_bytes = ctypes.c_int(0), ctypes.c_size_t(0)
    st = _lib.load().ia_conv2d_plan(b, i, 4 * o, h, w, 3, 0, 3, ctypes.byref(plan_s), ctypes.byref(plan_bytes))
    # whole rounds of the 128-channel x 256-point tile (a layer with fewer of those than CUs runs on the 32-channel tiles or stream-K)
    return st == 0 and plan_s.value == 0 and h * w >= 1024 and w <= 512 and b * ((h * w + 255) // 256) * (4 * o // 128) >= 256


def upconv_fir_sx(xs, wk, demod=None, noise=None, noise_strength=None, bias=None, styles_next=None, act='linear', alpha=0.2, gain=1.0,
                  clamp=None, want_f32=False, split_for=None, split_planes=2):
    """ia_upconv2d_fir_sx: an up-sampling layer (transposed convolution + resample FIR + noise + bias + activation) of a SplitAct as one
    stride-1 launch on the COMPOSED weight (pack of compose_upfir_weight).  Returns the SplitAct of the [B, O, 2H, 2W] result for
    `split_for` (multiplied by styles_next), or (y, SplitAct) with want_f32."""
    if not isinstance(xs, SplitAct):
        raise RuntimeError('xs must be a SplitAct')
    b, i, h, w = xs.shape
    o4 = wk.shape[-2]
    if o4 % 4 or wk.shape[-4] != 9 or wk.shape[-3] * 8 != i:
        raise RuntimeError(f'packed composed weight {tuple(wk.shape)} does not match 3x3, in-channels {i}, 4 x O rows')
    o = o4 // 4
    for name, t, n in (('demod', demod, b * o), ('noise', noise, 4 * h * w), ('bias', bias, o), ('styles_next', styles_next, b * o)):
        if t is not None and _f32c(t, name).numel() != n:
            raise RuntimeError(f'{name} has {t.numel()} elements, expected {n}')
    dev = xs.data.device
    y = torch.empty(b, o, 2 * h, 2 * w, device=dev, dtype=torch.float32) if want_f32 else None
    ys = torch.empty(b, split_planes, o // 8, 2 * h, 2 * w, 8, device=dev, dtype=torch.float16)
    flops = 2.0 * b * h * w * i * o * 9          # (algorithmic: the transposed convolution's; 4x of it is executed)
    traffic = 2.0 * (xs.data.numel() + wk.numel()) + 2.0 * ys.numel() + (4.0 * y.numel() if want_f32 else 0.0)
    with torch.cuda.device(dev), _Timed('conv2d_mfma_t', flops, traffic, f'B{b} I{i} O{o} {h}x{w} ' + ('f16x3 dma' if xs.planes == 2 else 'f16 dma') + ' composed up-FIR'):
        st = _lib.load().ia_upconv2d_fir_sx(_p(xs.data), int(xs.planes), _p(wk), int(getattr(wk, 'wk_exp', 0)), _p(demod), _p(noise),
                                            _p(noise_strength), _p(bias), _p(y), _p(ys), int(split_planes), _p(styles_next), b, i, o, h, w,
                                            ACT_ID[act], float(alpha), float(gain), float(-1 if clamp is None else clamp), _lib.stream_ptr(dev))
    _lib.check(st, 'ia_upconv2d_fir_sx')
    out = SplitAct(ys, o, split_for)
    return (y, out) if want_f32 else out


def planes_channels_last(planes):
    """[B, 3, C, H, W] -> [B, 3, H, W, C] contiguous: the layout ia_render_rays gathers from."""
    return planes.permute(0, 1, 3, 4, 2).contiguous()


def render_rays(planes_cl, rays_o, rays_d, jitter, dist, w0, b0, w1, b1, lr_multiplier=1.0, box_warp=1.0, white_back=False,
                n_coarse=48, n_importance=48, debug=False, channel_major=False, u_importance=None, split_styles=None, split_planes=0):
    """Fused importance renderer (see ia_render_rays).  Returns (rgb [B,R,32], depth [B,R,1], wsum [B,R,1][, aux]).
    channel_major=True stores rgb as [B,32,R] (the feature image the super-resolution head reads) and returns the [B,R,32] VIEW of
    it: same values and shape, and `rgb.permute(0, 2, 1).reshape(B, 32, nrr, nrr)` is then contiguous without a copy."""
    for name, t in (('planes', planes_cl), ('rays_o', rays_o), ('rays_d', rays_d), ('jitter', jitter), ('dist', dist),
                    ('w0', w0), ('b0', b0), ('w1', w1), ('b1', b1)):
        _f32c(t, name)
    b, three, ph, pw, c = planes_cl.shape
    if three != 3 or c != 32:
        raise RuntimeError(f'planes must be [B,3,H,W,32] channels-last, got {tuple(planes_cl.shape)}')
    r = rays_o.shape[1]
    if dist.numel() not in (1, b):
        raise RuntimeError(f'dist must have 1 or B = {b} elements, got {dist.numel()}')
    dist_per_frame = dist.numel() == b and b > 1
    if tuple(jitter.shape[:3]) != (b, r, n_coarse):
        raise RuntimeError(f'jitter must be [B,R,{n_coarse}(,1)], got {tuple(jitter.shape)}')
    if u_importance is not None and (_f32c(u_importance, 'u_importance').numel() != b * r * n_importance):
        raise RuntimeError(f'u_importance must hold B * R * {n_importance} sorted uniform draws')
    dev = planes_cl.device
    lib = _lib.load()
    rgb = torch.empty(b, 32, r, device=dev).permute(0, 2, 1) if channel_major else torch.empty(b, r, 32, device=dev)
    depth = torch.empty(b, r, 1, device=dev)
    wsum = torch.empty(b, r, 1, device=dev)
    scratch = torch.empty(2 * lib.ia_render_rays_grid(b, r) * (8 * b if dist_per_frame else 1), device=dev)      # (8 waves per workgroup: include/ia_hip.h)
    # split_planes = 1 | 2: the composited features ALSO in the split format (x split_styles [B,32]) as `rgb.split_data` [B, planes, 4, R, 8]
    split = torch.empty(b, split_planes, 4, r, 8, device=dev, dtype=torch.float16) if split_planes else None
    if split_styles is not None and (_f32c(split_styles, 'split_styles').numel() != b * 32 or not split_planes):
        raise RuntimeError('split_styles: [B,32] styles of the layer that consumes the split copy (with split_planes = 1 or 2)')
    aux = {}
    if debug:
        aux = dict(z_fine=torch.empty(b, r, 48, device=dev), inds=torch.empty(b, r, 48, device=dev, dtype=torch.int32),
                   order=torch.empty(b, r, 96, device=dev, dtype=torch.int32), w_coarse=torch.empty(b, r, 47, device=dev),
                   sigma_coarse=torch.empty(b, r, 48, device=dev))
    # algorithmic work: 2 passes x 48 samples x (32*64 + 64*33) MACs per ray; planes + rays in, 34 floats out
    flops = b * r * 96 * 2.0 * (32 * 64 + 64 * 33)
    traffic = 4.0 * (planes_cl.numel() + 2 * rays_o.numel() + jitter.numel() + b * r * 34)
    with torch.cuda.device(dev), _Timed('render_rays', flops, traffic):
        st = lib.ia_render_rays(_p(planes_cl), _p(rays_o), _p(rays_d), _p(jitter), _p(u_importance), _p(dist), _p(w0), _p(b0), _p(w1), _p(b1),
                                float(lr_multiplier), float(box_warp), int(bool(white_back)) | (2 if channel_major else 0) | (4 if dist_per_frame else 0), b, r, ph, pw,
                                int(n_coarse),
                                int(n_importance), _p(rgb), _p(depth), _p(wsum), _p(scratch), _p(aux.get('z_fine')),
                                _p(aux.get('inds')), _p(aux.get('order')), _p(aux.get('w_coarse')), _p(aux.get('sigma_coarse')),
                                _p(split), _p(split_styles), int(split_planes or 2), _lib.stream_ptr(dev))
    _lib.check(st, 'ia_render_rays')
    if split is not None:
        rgb.split_data = split
    return (rgb, depth, wsum, aux) if debug else (rgb, depth, wsum)


def ray_limits_box(rays_o, rays_d, box_side_length, repair_misses=False):
    """math_utils.get_ray_limits_box on the device (see ia_ray_limits_box) -> limits [..., 2] = (t_near, t_far); repair_misses: with
    ImportanceRenderer.forward's repair of the rays that miss the box (renderer.py:133-136), no host round trip."""
    _f32c(rays_o, 'rays_o'); _f32c(rays_d, 'rays_d')
    n = rays_o.numel() // 3
    lib = _lib.load()
    limits = torch.empty(*rays_o.shape[:-1], 2, device=rays_o.device)
    part = torch.empty(2 * lib.ia_ray_limits_box_parts(n), device=rays_o.device)
    with torch.cuda.device(rays_o.device):
        st = lib.ia_ray_limits_box(_p(rays_o), _p(rays_d), float(box_side_length), n, int(bool(repair_misses)), _p(limits), _p(part),
                                   _lib.stream_ptr(rays_o.device))
    _lib.check(st, 'ia_ray_limits_box')
    return limits


def render_rays_box(planes_cl, rays_o, rays_d, jitter, u_importance, w0, b0, w1, b1, ray_limits=None, ray_start=0.0, ray_end=0.0,
                    flip_z=False, lr_multiplier=1.0, box_warp=1.0, white_back=False, n_coarse=48, n_importance=48, debug=False):
    """ImportanceRenderer.forward as one launch (see ia_render_rays_box): `ray_limits` [B,R,2] (the 'auto' box limits) or the fixed
    [ray_start, ray_end]; `u_importance` [B*R, 48] sorted draws.  Returns (rgb [B,R,32], depth [B,R,1], wsum [B,R,1][, aux])."""
    for name, t in (('planes', planes_cl), ('rays_o', rays_o), ('rays_d', rays_d), ('jitter', jitter), ('u_importance', u_importance),
                    ('w0', w0), ('b0', b0), ('w1', w1), ('b1', b1)):
        _f32c(t, name)
    b, three, ph, pw, c = planes_cl.shape
    if three != 3 or c != 32:
        raise RuntimeError(f'planes must be [B,3,H,W,32] channels-last, got {tuple(planes_cl.shape)}')
    r = rays_o.shape[1]
    if tuple(jitter.shape[:3]) != (b, r, n_coarse):
        raise RuntimeError(f'jitter must be [B,R,{n_coarse}(,1)], got {tuple(jitter.shape)}')
    if u_importance.numel() != b * r * n_importance:
        raise RuntimeError(f'u_importance must hold B * R * {n_importance} sorted uniform draws')
    if ray_limits is not None and _f32c(ray_limits, 'ray_limits').numel() != b * r * 2:
        raise RuntimeError('ray_limits must be [B,R,2]')
    dev = planes_cl.device
    lib = _lib.load()
    rgb, depth, wsum = torch.empty(b, r, 32, device=dev), torch.empty(b, r, 1, device=dev), torch.empty(b, r, 1, device=dev)
    scratch = torch.empty(2 * lib.ia_render_rays_grid(b, r), device=dev)
    aux = {}
    if debug:
        aux = dict(z_fine=torch.empty(b, r, 48, device=dev), inds=torch.empty(b, r, 48, device=dev, dtype=torch.int32),
                   order=torch.empty(b, r, 96, device=dev, dtype=torch.int32), w_coarse=torch.empty(b, r, 47, device=dev),
                   sigma_coarse=torch.empty(b, r, 48, device=dev))
    flops = b * r * 96 * 2.0 * (32 * 64 + 64 * 33)
    traffic = 4.0 * (planes_cl.numel() + 2 * rays_o.numel() + jitter.numel() + u_importance.numel() + b * r * 36)
    with torch.cuda.device(dev), _Timed('render_rays', flops, traffic, 'box'):
        st = lib.ia_render_rays_box(_p(planes_cl), _p(rays_o), _p(rays_d), _p(jitter), _p(u_importance), _p(ray_limits), float(ray_start), float(ray_end),
                                    _p(w0), _p(b0), _p(w1), _p(b1), float(lr_multiplier), float(box_warp),
                                    int(bool(white_back)) | (8 if flip_z else 0), b, r, ph, pw, int(n_coarse), int(n_importance),
                                    _p(rgb), _p(depth), _p(wsum), _p(scratch), _p(aux.get('z_fine')), _p(aux.get('inds')), _p(aux.get('order')),
                                    _p(aux.get('w_coarse')), _p(aux.get('sigma_coarse')), _lib.stream_ptr(dev))
    _lib.check(st, 'ia_render_rays_box')
    return (rgb, depth, wsum, aux) if debug else (rgb, depth, wsum)


def importance_stage(z_coarse, w_coarse):
    """Importance resampling + merge order from given coarse depths/weights (parity-test entry)."""
    n = z_coarse.shape[0]
    dev = z_coarse.device
    z_fine = torch.empty(n, 48, device=dev)
    inds = torch.empty(n, 48, device=dev, dtype=torch.int32)
    order = torch.empty(n, 96, device=dev, dtype=torch.int32)
    st = _lib.load().ia_importance_stage(_p(_f32c(z_coarse, 'z_coarse')), _p(_f32c(w_coarse, 'w_coarse')), _p(z_fine), _p(inds),
                                         _p(order), n, _lib.stream_ptr(dev))
    _lib.check(st, 'ia_importance_stage')
    return z_fine, inds, order


def fill_mouth(alpha):
    """alpha [B,1,H,W] -> mouth mask [B,1,H,W] (see ia_fill_mouth)."""
    _f32c(alpha, 'alpha')
    b, one, h, w = alpha.shape
    mouth = torch.empty_like(alpha)
    with torch.cuda.device(alpha.device):
        st = _lib.load().ia_fill_mouth(_p(alpha), _p(mouth), b * one, h, w, _lib.stream_ptr(alpha.device))
    _lib.check(st, 'ia_fill_mouth')
    return mouth


def mouth_edge_blur(alpha, mouth):
    """alpha, mouth (= fill_mouth(alpha)) [B,1,H,W] -> the eroded + blurred mouth mask [B,1,H,W] (see ia_mouth_edge_blur)."""
    _f32c(alpha, 'alpha'); _f32c(mouth, 'mouth')
    b, one, h, w = alpha.shape
    out = torch.empty_like(alpha)
    with torch.cuda.device(alpha.device):
        st = _lib.load().ia_mouth_edge_blur(_p(alpha), _p(mouth), _p(out), b * one, h, w, _lib.stream_ptr(alpha.device))
    _lib.check(st, 'ia_mouth_edge_blur')
    return out


def rasterize_level(tex, uvcoords_image, upper_alpha, sta, bbox, res, tex_cl=None):
    """One level of TriPlaneGenerator.rasterize (see ia_rasterize_level).  tex [B,C,Rt,Rt]; sta NCHW (may be a channel
    slice of a wider tensor as long as channels/rows are dense); returns [B, C+1, res, res].  `tex_cl` is the level
    already in channels-last order (channels_last_copy), if the caller made it ahead of time."""
    b, c, rt, _ = tex.shape
    if tex_cl is None:
        tex_cl = channels_last_copy(tex)
    rs = sta.shape[-1]
    if not (sta.stride(3) == 1 and sta.stride(2) == rs and sta.stride(1) == rs * rs and sta.shape[1] >= c):
        sta = sta.contiguous()
    out = torch.empty(b, c + 1, res, res, device=tex.device, dtype=torch.float32)
    y0, y1, x0, x1 = bbox
    with torch.cuda.device(tex.device):
        st = _lib.load().ia_rasterize_level(_p(_f32c(tex_cl, 'tex')), _p(_f32c(uvcoords_image, 'uvcoords_image')),
                                            _p(_f32c(upper_alpha, 'upper_alpha')), sta.data_ptr(), sta.stride(0), _p(out),
                                            b, c, rt, rs, res, y0, y1, x0, x1, _lib.stream_ptr(tex.device))
    _lib.check(st, 'ia_rasterize_level')
    return out


def cond_blend(cond, x):
    """cond[:, :-1] * a + x * (1 - a) with a = cond[:, -1:] in one pass (see ia_cond_blend)."""
    b, c, h, w = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        st = _lib.load().ia_cond_blend(_p(_f32c(cond, 'cond')), _p(_f32c(x, 'x')), _p(y), b, c, h, w, _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_cond_blend')
    return y


def cond_blend_split(cond, x, styles_next, consumer):
    """cond_blend whose result exists only as the SplitAct `consumer` reads (see ia_cond_blend_split)."""
    b, c, h, w = x.shape
    ys = torch.empty(b, 2, c // 8, h, w, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device):
        st = _lib.load().ia_cond_blend_split(_p(_f32c(cond, 'cond')), _p(_f32c(x, 'x')), _p(None if styles_next is None else _f32c(styles_next, 'styles')),
                                             _p(ys), b, c, h, w, _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_cond_blend_split')
    return SplitAct(ys, c, consumer)


def channels_last_copy(t):
    """[B,C,H,W] -> contiguous [B,H,W,C] (the layout ia_rasterize_level gathers from): ia_channels_last for contiguous fp32 device
    tensors (a tiled transpose: 25 -> ~8 us for 256 channels @128^2 against the strided Tensor.copy_), torch otherwise."""
    if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 4:
        b, c, h, w = t.shape
        out = torch.empty(b, h, w, c, device=t.device, dtype=torch.float32)
        with torch.cuda.device(t.device):
            st = _lib.load().ia_channels_last(_p(t), _p(out), b, c, h, w, _lib.stream_ptr(t.device))
        _lib.check(st, 'ia_channels_last')
        return out
    return t.permute(0, 2, 3, 1).contiguous()


def blend_planes(stitch, full_alpha, static_planes, bbox):
    """Channels-last blended tri-planes [B,3,256,256,32] (see ia_blend_planes).  static_planes [B,96,256,256]."""
    b = stitch.shape[0]
    planes_cl = torch.empty(b, 3, 256, 256, 32, device=stitch.device, dtype=torch.float32)
    y0, y1, x0, x1 = bbox
    with torch.cuda.device(stitch.device):
        st = _lib.load().ia_blend_planes(_p(_f32c(stitch, 'stitch')), _p(_f32c(full_alpha, 'full_alpha')),
                                         _p(_f32c(static_planes, 'static_planes')), static_planes.stride(0), _p(planes_cl),
                                         b, y0, y1, x0, x1, _lib.stream_ptr(stitch.device))
    _lib.check(st, 'ia_blend_planes')
    return planes_cl


class StylePlan:
    """Device tables describing every modulated layer of one synthesis network (see ia_styles_demod).

    `layers` is a list of (module, w_index): module has .affine (FullyConnectedLayer), .weight and `demodulate`
    semantics given by `hasattr(module, 'noise_strength') or module is a SynthesisLayer` -> passed explicitly."""

    def __init__(self, entries, device):
        # entries: list of dict(affine=FC module, weight=conv weight param, widx=int, demod=bool)
        self.entries = entries
        self.device = device
        self.key = None
        self._build()

    def _signature(self):
        sig = []
        for e in self.entries:
            sig += [e['affine'].weight.data_ptr(), e['affine'].weight._version, e['weight'].data_ptr(), e['weight']._version]
        return tuple(sig)

    def _build(self):
        rows, soff, doff = [], 0, 0
        table, gains, srl, drl = [], [], [], []
        self.keep = []
        for li, e in enumerate(self.entries):
            fc = e['affine']
            a = fc.weight.detach().float().contiguous()
            bias = fc.bias.detach().float().contiguous() if fc.bias is not None else None
            i_dim = a.shape[0]
            wsq = weight_sq_sum(e['weight']) if e['demod'] else None
            o_dim = e['weight'].shape[0]
            self.keep += [a, bias, wsq]
            table.append([a.data_ptr(), bias.data_ptr() if bias is not None else 0, wsq.data_ptr() if wsq is not None else 0,
                          i_dim, o_dim, e['widx'], soff, doff if e['demod'] else -1])
            gains.append([float(fc.weight_gain), float(fc.bias_gain)])
            srl += [li] * i_dim
            e['soff'], e['I'] = soff, i_dim
            soff += i_dim
            if e['demod']:
                drl += [li] * o_dim
                e['doff'], e['O'] = doff, o_dim
                doff += o_dim
        dev = self.device
        self.table = torch.tensor(table, dtype=torch.int64, device=dev)
        self.gains = torch.tensor(gains, dtype=torch.float32, device=dev)
        self.srl = torch.tensor(srl, dtype=torch.int32, device=dev)
        self.drl = torch.tensor(drl if drl else [0], dtype=torch.int32, device=dev)
        self.srows, self.drows = soff, doff
        self.key = self._signature()

    def run(self, ws):
        """ws [B, num_ws, w_dim] -> list of (styles [B,I], demod [B,O] or None) per entry (views into two buffers)."""
        if self._signature() != self.key:
            self._build()
        ws = _f32c(ws.float().contiguous(), 'ws')
        b, num_ws, w_dim = ws.shape
        styles = torch.empty(b * self.srows, device=ws.device)
        demod = torch.empty(b * max(self.drows, 1), device=ws.device)
        with torch.cuda.device(ws.device):
            st = _lib.load().ia_styles_demod(_p(ws), b, num_ws, w_dim, _p(self.table), _p(self.gains), _p(self.srl), self.srows,
                                             _p(self.drl), self.drows, _p(styles), _p(demod), _lib.stream_ptr(ws.device))
        _lib.check(st, 'ia_styles_demod')
        self.last_buffers = (styles, demod)      # alive until the next run: the per-layer views may be consumed on other streams than this one
        out = []
        for e in self.entries:
            s = styles[b * e['soff']:b * (e['soff'] + e['I'])].view(b, e['I'])
            d = demod[b * e['doff']:b * (e['doff'] + e['O'])].view(b, e['O']) if e['demod'] else None
            out.append((s, d))
        return out


def ray_sampler(cam25, resolution, normalize=True):
    """cam25 [B,25] (cam2world 16 + intrinsics 9) -> rays_o, rays_d [B, res^2, 3] (see ia_ray_sampler)."""
    cam25 = _f32c(cam25.float().contiguous(), 'cam')
    b = cam25.shape[0]
    rays_o = torch.empty(b, resolution * resolution, 3, device=cam25.device)
    rays_d = torch.empty_like(rays_o)
    with torch.cuda.device(cam25.device):
        st = _lib.load().ia_ray_sampler(_p(cam25), cam25.stride(0), _p(rays_o), _p(rays_d), b, int(resolution), int(bool(normalize)),
                                        _lib.stream_ptr(cam25.device))
    _lib.check(st, 'ia_ray_sampler')
    return rays_o, rays_d


def convgru_gates(gates_pre, x, h):
    """cat[x, sigmoid(r_pre) * h] (see ia_convgru_gates)."""
    b, c, hh, w = x.shape
    xrh = torch.empty(b, 2 * c, hh, w, device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        st = _lib.load().ia_convgru_gates(_p(_f32c(gates_pre, 'gates_pre')), _p(_f32c(x, 'x')), _p(_f32c(h, 'h')), _p(xrh), b, c, hh, w,
                                          _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_convgru_gates')
    return xrh


def convgru_update(gates_pre, cand_pre, h, prelu_weight=None, x_next=None):
    """h' = (1 - z) h + z tanh(c_pre) and, with x_next, cat[x_next, h'] for the next step (see ia_convgru_update)."""
    b, c, hh, w = h.shape
    h_out = torch.empty_like(h)
    xh = torch.empty(b, 2 * c, hh, w, device=h.device, dtype=torch.float32) if x_next is not None else None
    with torch.cuda.device(h.device):
        st = _lib.load().ia_convgru_update(_p(_f32c(gates_pre, 'gates_pre')), _p(_f32c(cand_pre, 'cand_pre')), _p(_f32c(h, 'h')),
                                           _p(None if prelu_weight is None else _f32c(prelu_weight, 'prelu_weight')), _p(h_out),
                                           _p(None if x_next is None else _f32c(x_next, 'x_next')), _p(xh), b, c, hh, w, _lib.stream_ptr(h.device))
    _lib.check(st, 'ia_convgru_update')
    return h_out, xh



def convgru_gates_split(gates_pre, x, h, consumer=None):
    """SplitAct of cat[x, sigmoid(r_pre) * h] (see ia_convgru_gates_split): the input of the cell's second convolution."""
    b, c, hh, w = x.shape
    out = torch.empty(b, 2, 2 * c // 8, hh, w, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device):
        st = _lib.load().ia_convgru_gates_split(_p(_f32c(gates_pre, 'gates_pre')), _p(_f32c(x, 'x')), _p(_f32c(h, 'h')), _p(out), b, c, hh, w,
                                                _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_convgru_gates_split')
    return SplitAct(out, 2 * c, consumer)


def convgru_update_split(gates_pre, cand_pre, h, prelu_weight=None, x_next=None, consumer=None):
    """h' (fp32) and, with x_next, the SplitAct of cat[x_next, h'] for the next step's first convolution (see ia_convgru_update_split)."""
    b, c, hh, w = h.shape
    h_out = torch.empty_like(h)
    xh = torch.empty(b, 2, 2 * c // 8, hh, w, 8, device=h.device, dtype=torch.float16) if x_next is not None else None
    with torch.cuda.device(h.device):
        st = _lib.load().ia_convgru_update_split(_p(_f32c(gates_pre, 'gates_pre')), _p(_f32c(cand_pre, 'cand_pre')), _p(_f32c(h, 'h')),
                                                 _p(None if prelu_weight is None else _f32c(prelu_weight, 'prelu_weight')), _p(h_out),
                                                 _p(None if x_next is None else _f32c(x_next, 'x_next')), _p(xh), b, c, hh, w, _lib.stream_ptr(h.device))
    _lib.check(st, 'ia_convgru_update_split')
    return h_out, (SplitAct(xh, 2 * c, consumer) if xh is not None else None)


def stage_inputs(pairs):
    """Copy each (src, dst) pair of same-shaped contiguous device tensors in ONE launch (ia_stage_inputs): the per-frame inputs
    of a captured frame into the graph's static buffers.  Pairs the kernel cannot take (broadcast / strided / other dtype sources)
    fall through to Tensor.copy_."""
    import ctypes
    segs = []
    for src, dst in pairs:
        if (src.is_cuda and src.device == dst.device and src.dtype == dst.dtype and src.shape == dst.shape
                and src.is_contiguous() and dst.is_contiguous() and src.numel() > 0):
            segs.append((src, dst))
        else:
            dst.copy_(src)
    for i in range(0, len(segs), 8):
        part = segs[i:i + 8]
        n = len(part)
        srcs = (ctypes.c_void_p * n)(*[s_.data_ptr() for s_, _ in part])
        dsts = (ctypes.c_void_p * n)(*[d_.data_ptr() for _, d_ in part])
        nbytes = (ctypes.c_int64 * n)(*[s_.numel() * s_.element_size() for s_, _ in part])
        dev = part[0][1].device
        with torch.cuda.device(dev):
            st = _lib.load().ia_stage_inputs(srcs, dsts, nbytes, n, _lib.stream_ptr(dev))
        _lib.check(st, 'ia_stage_inputs')


def attention(q, kv, heads, scale):
    """softmax(Q K^T * scale) V per head in one launch (see ia_attention).  q [B,N,C], kv [B,M,2C] (k = kv[..., :C], v = kv[..., C:]);
    returns [B,N,C]."""
    _f32c(q, 'q')
    _f32c(kv, 'kv')
    b, n, c = q.shape
    m = kv.shape[1]
    if kv.shape[0] != b or kv.shape[2] != 2 * c or c % heads:
        raise RuntimeError(f'attention: q {tuple(q.shape)} / kv {tuple(kv.shape)} / {heads} heads do not fit together')
    out = torch.empty_like(q)
    k, v = kv[..., :c], kv[..., c:]
    with torch.cuda.device(q.device), _Timed('attention', 4.0 * b * n * m * c, 4.0 * (q.numel() + kv.numel() + out.numel()), f'B{b} N{n} M{m} C{c}'):
        st = _lib.load().ia_attention(_p(q), k.data_ptr(), v.data_ptr(), _p(out), b, heads, n, m, c // heads, q.stride(0), q.stride(1),
                                      kv.stride(0), kv.stride(1), kv.stride(0), kv.stride(1), out.stride(0), out.stride(1), float(scale),
                                      _lib.stream_ptr(q.device))
    _lib.check(st, 'ia_attention')
    return out


class SplitTokens:
    """A token matrix [M, K] stored as fp16 hi / lo pairs [2, K/8, M, 8] (ia_tokens_split): the operand of linear_sx."""

    def __init__(self, data, rows, cols, lead_shape):
        self.data, self.rows, self.cols, self.lead_shape = data, rows, cols, tuple(lead_shape)


def linear_supported(in_features):
    return in_features % 16 == 0


def pack_linear_weight_split(w):
    """nn.Linear weight [N, K] -> the convolution weight format of a 1x1 kernel, fp16 [2, 1, K/8, N, 8] with its scale as `.wk_exp`
    (pack_conv_weight_split): the weight operand of linear_sx."""
    n, k = w.shape
    return pack_conv_weight_split(w.detach().reshape(n, k, 1, 1))


def tokens_split(x):
    """fp32 [..., K] (rows contiguous, or a column slice of such a tensor: rows x.stride(-2) apart) -> SplitTokens (see ia_tokens_split).
    One split serves every linear layer that reads x."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.stride(-1) == 1):
        raise RuntimeError('tokens_split: x must be a float32 device tensor with contiguous rows')
    k = x.shape[-1]
    m = x.numel() // k
    ld = k
    if not x.is_contiguous():
        ld = x.stride(-2)
        if x.dim() < 2 or any(x.stride(d) != x.stride(d + 1) * x.shape[d + 1] for d in range(x.dim() - 2)):
            raise RuntimeError('tokens_split: rows must be evenly spaced')
    out = torch.empty(2, k // 8, m, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device), _Timed('tokens_split', 0.0, 8.0 * x.numel(), f'M{m} K{k}'):
        st = _lib.load().ia_tokens_split(_p(x), int(ld), _p(out), m, k, _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_tokens_split')
    return SplitTokens(out, m, k, x.shape[:-1])


def layernorm_split_supported(k):
    return k in (512, 1024, 2048)


def layernorm_split(x, norm):
    """nn.LayerNorm `norm` over the last dimension of x [..., K] + tokens_split of the result, one launch (see ia_layernorm_split)."""
    _f32c(x, 'x')
    k = x.shape[-1]
    m = x.numel() // k
    out = torch.empty(2, k // 8, m, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device), _Timed('layernorm_split', 8.0 * x.numel(), 8.0 * x.numel(), f'M{m} K{k}'):
        st = _lib.load().ia_layernorm_split(_p(x), _p(_f32c(norm.weight.detach(), 'weight')), _p(_f32c(norm.bias.detach(), 'bias')), float(norm.eps),
                                            _p(out), m, k, _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_layernorm_split')
    return SplitTokens(out, m, k, x.shape[:-1])


ATTENTION_SX_ONE_LAUNCH = True      # head_dim 256: ia_attention_sx (online softmax) instead of matmul / softmax / matmul


def attention_sx_supported(head_dim, n, m):
    return head_dim % 16 == 0 and m % 16 == 0 and m <= 4096


def attention_sx(q, kv, heads, scale):
    """softmax(Q K^T * scale) V per head through the fp16-pair GEMM (see ia_matmul_sx): q [B, N, C], kv [B, M, 2C] (k = kv[..., :C],
    v = kv[..., C:]); returns [B, N, C].  The score matrix [heads, N, M] passes through HBM once (fp32) and once as fp16 pairs."""
    _f32c(q, 'q')
    _f32c(kv, 'kv')
    b, n, c = q.shape
    m = kv.shape[1]
    hd = c // heads
    if kv.shape[0] != b or kv.shape[2] != 2 * c or c % heads or not attention_sx_supported(hd, n, m):
        raise RuntimeError(f'attention_sx: q {tuple(q.shape)} / kv {tuple(kv.shape)} / {heads} heads are not covered')
    lib, dev = _lib.load(), q.device
    out = torch.empty_like(q)
    with torch.cuda.device(dev), _Timed('attention_sx', 4.0 * b * n * m * c, 4.0 * (q.numel() + kv.numel() + out.numel()) + 16.0 * b * heads * n * m,
                                        f'B{b} N{n} M{m} C{c}'):
        s_ = _lib.stream_ptr(dev)
        scores = torch.empty(heads, n, m, device=dev, dtype=torch.float32)
        probs = torch.empty(2, heads, m // 8, n, 8, device=dev, dtype=torch.float16)
        vt = torch.empty(2, m // 8, c, 8, device=dev, dtype=torch.float16)
        for i in range(b):
            qs, ks = tokens_split(q[i]), tokens_split(kv[i, :, :c])
            flash = ATTENTION_SX_ONE_LAUNCH and bool(lib.ia_attention_sx_supported(hd, n, m))
            _lib.check(lib.ia_tokens_split_t(kv[i, :, c:].data_ptr(), kv.stride(1), _p(vt), m, c, int(flash), s_), 'ia_tokens_split_t')
            if flash:        # one launch, no score matrix (head_dim 256)
                _lib.check(lib.ia_attention_sx(_p(qs.data), _p(ks.data), _p(vt), _p(out[i]), heads, n, m, hd, float(scale), s_), 'ia_attention_sx')
                continue
            _lib.check(lib.ia_matmul_sx(_p(qs.data), _p(ks.data), _p(scores), heads, n, m, hd, n, (c // 8) * n * 16, (hd // 8) * n * 16,
                                        m, (c // 8) * m * 16, (hd // 8) * m * 16, n * m, m, float(scale), s_), 'ia_matmul_sx')
            _lib.check(lib.ia_softmax_split(_p(scores), _p(probs), heads, n, m, s_), 'ia_softmax_split')
            _lib.check(lib.ia_matmul_sx(_p(probs), _p(vt), _p(out[i]), heads, n, hd, m, n, heads * (m // 8) * n * 16, (m // 8) * n * 16,
                                        c, (m // 8) * c * 16, hd * 16, hd, c, 1.0, s_), 'ia_matmul_sx')
    return out


def pack_patch_weight_split(w):
    """Conv2d weight [N, C, k, k] -> pack_linear_weight_split of its [N, C*k*k] matrix, columns zero-padded to a multiple of 16: the
    weight operand of linear_sx over im2col_split tokens."""
    n = w.shape[0]
    w2 = w.detach().reshape(n, -1)
    kp = (w2.shape[1] + 15) // 16 * 16
    if kp != w2.shape[1]:
        w2 = torch.nn.functional.pad(w2, (0, kp - w2.shape[1]))
    return pack_linear_weight_split(w2.contiguous())


def im2col_split(x, ksize, stride, pad):
    """NCHW fp32 image -> SplitTokens of its ksize x ksize patches (see ia_im2col_split); lead shape (B, OH * OW)."""
    _f32c(x, 'x')
    b, c, h, w = x.shape
    oh, ow = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    kp = (c * ksize * ksize + 15) // 16 * 16
    m = b * oh * ow
    out = torch.empty(2, kp // 8, m, 8, device=x.device, dtype=torch.float16)
    with torch.cuda.device(x.device), _Timed('im2col_split', 0.0, 4.0 * x.numel() + 4.0 * m * kp, f'B{b} C{c} {h}x{w} k{ksize} s{stride}'):
        st = _lib.load().ia_im2col_split(_p(x), _p(out), b, c, h, w, int(ksize), int(stride), int(pad), _lib.stream_ptr(x.device))
    _lib.check(st, 'ia_im2col_split')
    t = SplitTokens(out, m, kp, (b, oh * ow))
    t.grid = (oh, ow)
    return t


def linear_sx_splitk(xs, w_split, bias=None):
    """linear_sx with K cut over the launch when the library's plan says so (few rows, very long K; see ia_linear_sx_splitk)."""
    k8, n = w_split.shape[2], w_split.shape[3]
    ks, nbytes = ctypes.c_int(0), ctypes.c_size_t(0)
    _lib.check(_lib.load().ia_linear_splitk_plan(xs.rows, xs.cols, n, ctypes.byref(ks), ctypes.byref(nbytes)), 'ia_linear_splitk_plan')
    if ks.value <= 1:
        return linear_sx(xs, w_split, bias)
    if k8 * 8 != xs.cols:
        raise RuntimeError(f'linear_sx_splitk: {xs.cols} input features against a weight of {k8 * 8}')
    y = torch.empty(*xs.lead_shape, n, device=xs.data.device, dtype=torch.float32)
    scratch = torch.empty(nbytes.value // 4, device=y.device, dtype=torch.float32)
    with torch.cuda.device(y.device), _Timed('linear_sx', 2.0 * xs.rows * xs.cols * n, 4.0 * (xs.rows * xs.cols + xs.cols * n + y.numel()),
                                             f'M{xs.rows} K{xs.cols} N{n} splitk{ks.value}'):
        st = _lib.load().ia_linear_sx_splitk(_p(xs.data), _p(w_split), int(w_split.wk_exp), _p(None if bias is None else _f32c(bias, 'bias')), _p(y),
                                             xs.rows, xs.cols, n, ks.value, _p(scratch), nbytes.value, _lib.stream_ptr(y.device))
    _lib.check(st, 'ia_linear_sx_splitk')
    return y


def linear_sx(xs, w_split, bias=None, residual=None, gelu=False):
    """act(x @ w^T + bias) + residual on the fp16-pair GEMM (see ia_linear_sx).  xs: SplitTokens; w_split: pack_linear_weight_split(w);
    residual: fp32 [..., N] like the output.  Returns fp32 [*lead, N]."""
    if not (w_split.dtype == torch.float16 and w_split.dim() == 5 and w_split.shape[0] == 2 and w_split.shape[1] == 1 and hasattr(w_split, 'wk_exp')):
        raise RuntimeError('linear_sx: the weight must come from pack_linear_weight_split')
    k8, n = w_split.shape[2], w_split.shape[3]
    if k8 * 8 != xs.cols:
        raise RuntimeError(f'linear_sx: {xs.cols} input features against a weight of {k8 * 8}')
    y = torch.empty(*xs.lead_shape, n, device=xs.data.device, dtype=torch.float32)
    if residual is not None:
        _f32c(residual, 'residual')
        if residual.shape != y.shape:
            raise RuntimeError(f'linear_sx: residual {tuple(residual.shape)} against an output of {tuple(y.shape)}')
    with torch.cuda.device(y.device), _Timed('linear_sx', 2.0 * xs.rows * xs.cols * n, 4.0 * (xs.rows * xs.cols + xs.cols * n + y.numel()),
                                             f'M{xs.rows} K{xs.cols} N{n}'):
        st = _lib.load().ia_linear_sx(_p(xs.data), _p(w_split), int(w_split.wk_exp), _p(None if bias is None else _f32c(bias, 'bias')),
                                      _p(residual), _p(y), xs.rows, xs.cols, n, int(bool(gelu)), _lib.stream_ptr(y.device))
    _lib.check(st, 'ia_linear_sx')
    return y


def se_gate_split(v, shortcut, w1, w2, next_scale=None, next_shift=None, consumer=None):
    """ia_se_gate_split: (out, SplitAct of out * next_scale + next_shift or None).  next_scale / next_shift: [B, C] rows (the eval-mode
    BatchNorm in front of the convolution that reads `out` next) or both None."""
    b, c, h, w = v.shape
    if not (v.is_cuda and v.dtype == torch.float32 and shortcut.dtype == torch.float32 and tuple(shortcut.shape) == (b, c, h, w)):
        raise RuntimeError('se_gate_split: v and shortcut must be float32 device tensors of one shape')
    if (next_scale is None) != (next_shift is None):
        raise RuntimeError('next_scale and next_shift come together')
    r = w1.shape[0]
    _f32c(w1, 'w1'); _f32c(w2, 'w2')
    if next_scale is not None and (_f32c(next_scale, 'next_scale').numel() != b * c or _f32c(next_shift, 'next_shift').numel() != b * c):
        raise RuntimeError(f'next_scale / next_shift must have {b} x {c} elements')
    out = torch.empty(b, c, h, w, device=v.device, dtype=torch.float32)
    pooled = torch.empty(b * c, device=v.device, dtype=torch.float32)
    ys = torch.empty(b, 2, c // 8, h, w, 8, device=v.device, dtype=torch.float16) if next_scale is not None else None
    with torch.cuda.device(v.device):
        st = _lib.load().ia_se_gate_split(v.data_ptr(), _lib.strides64(v), shortcut.data_ptr(), _lib.strides64(shortcut), _p(w1), _p(w2), _p(pooled),
                                          _p(out), _p(next_scale), _p(next_shift), _p(ys), b, c, r, h, w, _lib.stream_ptr(v.device))
    _lib.check(st, 'ia_se_gate_split')
    return out, (SplitAct(ys, c, consumer) if ys is not None else None)


def se_gate(v, shortcut, w1, w2):
    """v * sigmoid(w2 relu(w1 mean_hw(v))) + shortcut (see ia_se_gate).  v, shortcut: fp32 [B,C,H,W] views (any strides);
    w1 [R,C], w2 [C,R]."""
    b, c, h, w = v.shape
    if not (v.is_cuda and v.dtype == torch.float32 and shortcut.dtype == torch.float32 and tuple(shortcut.shape) == (b, c, h, w)):
        raise RuntimeError('se_gate: v and shortcut must be float32 device tensors of one shape')
    r = w1.shape[0]
    _f32c(w1, 'w1'); _f32c(w2, 'w2')
    out = torch.empty(b, c, h, w, device=v.device, dtype=torch.float32)
    pooled = torch.empty(b * c, device=v.device, dtype=torch.float32)
    with torch.cuda.device(v.device):
        st = _lib.load().ia_se_gate(v.data_ptr(), _lib.strides64(v), shortcut.data_ptr(), _lib.strides64(shortcut), _p(w1), _p(w2), _p(pooled),
                                    _p(out), b, c, r, h, w, _lib.stream_ptr(v.device))
    _lib.check(st, 'ia_se_gate')
    return out
