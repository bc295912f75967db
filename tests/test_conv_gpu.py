"""Parity of the fp32 MFMA convolution (ia_conv2d_mfma + ia_upfirdn2d_bias_act + ia_modconv_demod)
against the oracle's modulated_conv2d / synthesis-layer arithmetic."""
import pytest
import torch

from oracle import ops as O
from invertavatar_amd import hipops
from conftest import rnd, max_abs

pytestmark = pytest.mark.gpu


def _layer_ref(x, w, styles, noise, ns, bias, up, gain=1.0, clamp=None, demodulate=True, act='lrelu'):
    f = O.setup_filter([1, 3, 3, 1])
    pad = w.shape[-1] // 2
    y = O.modulated_conv2d(x, w, styles, noise=None if noise is None else noise * ns, up=up, padding=pad,
                           resample_filter=f, demodulate=demodulate)
    g = (O.SQRT2 if act == 'lrelu' else 1.0) * gain
    return O.bias_act(y, bias, act=act, gain=g, clamp=None if clamp is None else clamp * gain)


def _layer_hip(x, w, styles, noise, ns, bias, up, gain=1.0, clamp=None, demodulate=True, act='lrelu', ksplit=None, residual=None):
    dev = 'cuda'
    xd, sd = x.to(dev), styles.to(dev)
    wk = hipops.pack_conv_weight(w.to(dev))
    d = hipops.modconv_demod(sd, hipops.weight_sq_sum(w.to(dev))) if demodulate else None
    nz = None if noise is None else noise.to(dev).contiguous()
    nsd = None if noise is None else torch.tensor(ns, device=dev)
    bd = None if bias is None else bias.to(dev)
    g = (O.SQRT2 if act == 'lrelu' else 1.0) * gain
    cl = None if clamp is None else clamp * gain
    if up == 1:
        return hipops.conv2d_mfma(xd, wk, sd, d, nz, nsd, bd, residual, ksize=w.shape[-1], act=act, gain=g, clamp=cl, ksplit=ksplit)
    t = hipops.conv2d_mfma(xd, wk, sd, d, ksize=3, transposed=True, ksplit=ksplit)
    f = O.setup_filter([1, 3, 3, 1]).to(dev)
    h, w_ = x.shape[2] * 2, x.shape[3] * 2
    return hipops.upfirdn2d_bias_act(t, f, nz, nsd, bd, up=1, pad0=(1, 1), out_hw=(h, w_), fir_gain=4.0, act=act, act_gain=g, clamp=cl)


CASES = [  # B, I, O, res, up
    (1, 16, 32, 4, 1), (2, 24, 40, 8, 1), (1, 64, 128, 16, 1), (1, 32, 32, 32, 1), (2, 16, 16, 64, 1),
    (1, 8, 136, 128, 1), (1, 32, 96, 33, 1),
    (1, 16, 32, 4, 2), (2, 24, 40, 8, 2), (1, 32, 64, 16, 2), (1, 16, 16, 64, 2), (1, 8, 72, 37, 2),
    (1, 16, 10, 16, 2), (2, 8, 6, 21, 2), (1, 16, 35, 32, 2),      # transposed form, channel counts that are not multiples of 4
]


@pytest.mark.parametrize('b,i,o,res,up', CASES)
def test_synthesis_layer(b, i, o, res, up):
    x = rnd(1, b, i, res, res)
    w = rnd(2, o, i, 3, 3)
    styles = rnd(3, b, i) * 0.3 + 1
    out = res * up
    noise, bias = rnd(4, out, out), rnd(5, o) * 0.2
    ref = _layer_ref(x, w, styles, noise, 0.1, bias, up)
    got = _layer_hip(x, w, styles, noise, 0.1, bias, up).cpu()
    scale = ref.abs().max().item()
    assert got.shape == ref.shape
    assert max_abs(got, ref) <= 3e-5 * max(scale, 1.0), (max_abs(got, ref), scale)


@pytest.mark.parametrize('ksplit', [1, 2, 4, 8, 24, 64])
def test_split_k_matches(ksplit):
    x, w, styles = rnd(1, 1, 64, 16, 16), rnd(2, 48, 64, 3, 3), rnd(3, 1, 64) * 0.3 + 1
    ref = _layer_ref(x, w, styles, None, 0, rnd(5, 48), 1, clamp=1.5)
    got = _layer_hip(x, w, styles, None, 0, rnd(5, 48), 1, clamp=1.5, ksplit=ksplit).cpu()
    assert max_abs(got, ref) <= 3e-5
    ref = _layer_ref(x, w, styles, None, 0, rnd(5, 48), 2)
    got = _layer_hip(x, w, styles, None, 0, rnd(5, 48), 2, ksplit=ksplit).cpu()
    assert max_abs(got, ref) <= 5e-5


@pytest.mark.parametrize('transposed', [False, True])
def test_whole_rounds_plus_stream_k_leftovers(transposed):
    """B = 4 leaves 128 workgroup slots per batch element; 130 tiles = one whole round + 2 stream-K leftovers.
    Checked against the device library convolution, twice (the flag header must come back zeroed), bit-identical."""
    h, w, o = (127, 129, 64) if transposed else (128, 130, 128)
    x = torch.randn(4, 16, h, w, device='cuda')
    wt = torch.randn(o, 16, 3, 3, device='cuda') * 0.1
    wk = hipops.pack_conv_weight(wt)
    if transposed:
        ref = torch.nn.functional.conv_transpose2d(x, wt.transpose(0, 1), stride=2)
    else:
        ref = torch.nn.functional.conv2d(x, wt, padding=1)
    got = hipops.conv2d_mfma(x, wk, ksize=3, transposed=transposed)
    again = hipops.conv2d_mfma(x, wk, ksize=3, transposed=transposed)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 5e-4          # library conv accumulates in a different order
    assert torch.equal(got, again)


def test_stream_k_is_deterministic_under_concurrency():
    """The same layer on two streams at once (separate scratch per stream) and repeatedly: always the same bits."""
    x = torch.randn(1, 256, 32, 32, device='cuda')
    wk = hipops.pack_conv_weight(torch.randn(256, 256, 3, 3, device='cuda') * 0.05)
    first = hipops.conv2d_mfma(x, wk, ksize=3)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(4):
        for s in (s1, s2):
            with torch.cuda.stream(s):
                outs.append(hipops.conv2d_mfma(x, wk, ksize=3))
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x, wk.reshape(3, 3, 256, 256).permute(3, 2, 0, 1), padding=1)
    assert (first - ref).abs().max().item() <= 2e-3
    assert all(torch.equal(first, o) for o in outs)


@pytest.mark.parametrize('o', [3, 32, 96])
def test_torgb_with_skip(o):
    """1x1 modulated conv without demodulation + bias + residual skip image (ToRGB + img.add_)."""
    x, w, styles = rnd(1, 2, 48, 32, 32), rnd(2, o, 48, 1, 1), rnd(3, 2, 48)
    bias, skip = rnd(4, o), rnd(5, 2, o, 32, 32)
    ref = _layer_ref(x, w, styles, None, 0, bias, 1, demodulate=False, act='linear') + skip
    got = _layer_hip(x, w, styles, None, 0, bias, 1, demodulate=False, act='linear', residual=skip.cuda()).cpu()
    assert max_abs(got, ref) <= 3e-5


def test_conv_linearity_at_full_size():
    """Size-independent property at a BASELINE-sized layer (128 -> 128 channels @ 256^2): linear in x."""
    w = torch.randn(128, 128, 3, 3, device='cuda') * 0.05
    wk = hipops.pack_conv_weight(w)
    a, b = torch.randn(1, 128, 256, 256, device='cuda'), torch.randn(1, 128, 256, 256, device='cuda')
    f = lambda t: hipops.conv2d_mfma(t, wk, ksize=3)
    assert (f(a + 2 * b) - (f(a) + 2 * f(b))).abs().max().item() <= 1e-3
    # and against the library convolution on device (cross-check of tap orientation at full size)
    ref = torch.nn.functional.conv2d(a, w, padding=1)
    assert (f(a) - ref).abs().max().item() <= 2e-3
    reft = torch.nn.functional.conv_transpose2d(a, w.transpose(0, 1), stride=2)
    got = hipops.conv2d_mfma(a, wk, ksize=3, transposed=True)
    assert got.shape == reft.shape and (got - reft).abs().max().item() <= 2e-3


@pytest.mark.parametrize('transposed', [False, True])
def test_fp16_operand_form_is_exact_on_fp16_rounded_operands(transposed):
    """ia_conv2d_mfma_h rounds the style-scaled input and the weights to fp16 and accumulates in fp32: on operands that are
    already fp16 numbers it must agree with the fp32 library convolution up to summation order."""
    i, o, h, w = (16, 64, 33, 33) if transposed else (32, 128, 64, 64)
    x = (torch.randn(2, i, h, w, device='cuda')).half().float()
    wt = (torch.randn(o, i, 3, 3, device='cuda') * 0.1).half().float()
    wk_h = hipops.pack_conv_weight_h(wt)
    assert hipops.conv_h_supported(i, o, h, w, 3, transposed)
    got = hipops.conv2d_mfma(x, wk_h, ksize=3, transposed=transposed)
    if transposed:
        ref = torch.nn.functional.conv_transpose2d(x, wt.transpose(0, 1), stride=2)
    else:
        ref = torch.nn.functional.conv2d(x, wt, padding=1)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-4
    # and with styles: x * s is rounded to fp16 before the product (relative 2^-11 per term)
    s = torch.rand(2, i, device='cuda') + 0.5
    got = hipops.conv2d_mfma(x, wk_h, styles=s, ksize=3, transposed=transposed)
    xs = (x * s[:, :, None, None]).half().float()
    ref = (torch.nn.functional.conv_transpose2d(xs, wt.transpose(0, 1), stride=2) if transposed
           else torch.nn.functional.conv2d(xs, wt, padding=1))
    assert (got - ref).abs().max().item() <= 1e-3    # a tie in the fp16 rounding of one x*s flips an operand by one fp16 ulp


@pytest.mark.parametrize('transposed', [False, True])
def test_split_fp16_form_is_an_fp32_convolution(transposed):
    """ia_conv2d_mfma_s (hi + lo fp16 pairs, three products, fp32 accumulation) against an fp64 convolution: its error must
    be of the size of the fp32 MFMA form's own error (both are dominated by fp32 accumulation), far below the fp16 form's."""
    i, o, h, w = (64, 64, 65, 65) if transposed else (128, 128, 64, 64)
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(1, i, h, w, device='cuda', generator=g) * 3
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g)
    s = torch.rand(1, i, device='cuda', generator=g) + 0.5
    xs = (x * s[:, :, None, None]).double()
    ref = (torch.nn.functional.conv_transpose2d(xs, wt.double().transpose(0, 1), stride=2) if transposed
           else torch.nn.functional.conv2d(xs, wt.double(), padding=1))
    scale = ref.abs().max().item()
    err, rms = {}, {}
    for name, wk in (('f32', hipops.pack_conv_weight(wt)), ('split', hipops.pack_conv_weight_split(wt)), ('f16', hipops.pack_conv_weight_h(wt))):
        got = hipops.conv2d_mfma(x, wk, styles=s, ksize=3, transposed=transposed)
        err[name] = (got.double() - ref).abs().max().item() / scale
        rms[name] = (got.double() - ref).square().mean().sqrt().item() / scale
    print(f'relative max error vs fp64: {err}; rms: {rms}')
    assert err['f32'] <= 2e-6
    # the pair form keeps 22 operand bits and rounds once per 16-deep MFMA instead of once per product, so it is
    # at least as accurate as the fp32 MFMA chain, in the maximum and in the mean
    assert err['split'] <= err['f32'] and rms['split'] <= rms['f32']
    assert err['f16'] >= 20 * err['split']


@pytest.mark.parametrize('wscale', [1e-5, 1.0, 3e3])
def test_split_fp16_form_over_the_operand_range(wscale):
    """The fp16-pair form keeps every factor a normal fp16 number by powers of two (weights packed at 2^wk_exp, low parts of
    the activations at 2^11), so its accuracy must not depend on the scale of the weights, and input channels whose
    style-scaled activations fall below fp16's normal range (2^-14) must still contribute (they ride in the low part)."""
    i, o, h, w = 128, 128, 64, 64
    g = torch.Generator(device='cuda').manual_seed(11)
    x = torch.randn(1, i, h, w, device='cuda', generator=g) * 3
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g) * wscale
    s = torch.rand(1, i, device='cuda', generator=g) + 0.5
    s[:, ::2] *= 2e-5                                                   # every other channel: activations ~ 6e-5 and below
    xs = (x * s[:, :, None, None]).double()
    ref = torch.nn.functional.conv2d(xs, wt.double(), padding=1)
    tiny_only = torch.nn.functional.conv2d(xs[:, ::2], wt.double()[:, ::2], padding=1)
    scale = ref.abs().max().item()
    got = hipops.conv2d_mfma(x, hipops.pack_conv_weight_split(wt), styles=s, ksize=3)
    err = (got.double() - ref).abs().max().item() / scale
    share = tiny_only.abs().max().item() / scale
    print(f'weights x {wscale:g}: relative max error vs fp64 {err:.2e}; the tiny channels alone are {share:.1e} of the output')
    assert share > 15 * 1e-6      # (dropping them would show)
    assert err <= 1e-6


def test_split_fp16_form_large_transposed_tile():
    """Point grids of >= 32768 points (the 256^2 -> 512^2 layers) take the 64ch x 128pt x 4-phase tile of the fp16-pair form:
    same arithmetic, two point fragments per wave; checked against an fp64 transposed convolution on an odd-sized image whose
    tiles straddle rows, with stream-K leftovers (B = 2)."""
    i, o, h, w = 32, 64, 201, 187
    g = torch.Generator(device='cuda').manual_seed(13)
    x = torch.randn(2, i, h, w, device='cuda', generator=g) * 2
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g)
    s = torch.rand(2, i, device='cuda', generator=g) + 0.5
    ref = torch.nn.functional.conv_transpose2d((x * s[:, :, None, None]).double(), wt.double().transpose(0, 1), stride=2)
    got = hipops.conv2d_mfma(x, hipops.pack_conv_weight_split(wt), styles=s, ksize=3, transposed=True)
    assert got.shape == ref.shape
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f'large transposed fp16-pair tile: relative max error vs fp64 {err:.2e}')
    assert err <= 1e-6


# ---------------------------------------------------------------- split-format (LDS-DMA) convolution: ia_act_split + ia_conv2d_mfma_sx
def _split_reference(x, s):
    """The fp16 pair the kernels must store: hi = fp16(v) (0 below the normal range), lo = fp16((v - hi) * 2^11), v = x * s saturated."""
    v = (x * s[:, :, None, None]).clamp(-65504.0, 65504.0)
    hi = torch.where(v.abs() < 6.103515625e-5, torch.zeros_like(v), v).half()
    lo = ((v - hi.float()) * 2048.0).half()
    return hi, lo


def test_act_split_format():
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(2, 24, 9, 13, device='cuda', generator=g) * torch.logspace(-7, 4, 24, device='cuda')[None, :, None, None]
    x[0, 0, 0, 0], x[0, 1, 0, 0] = 1e6, -1e6                 # saturate
    s = torch.rand(2, 24, device='cuda', generator=g) + 0.5
    xs = hipops.act_split(x, s)
    hi, lo = _split_reference(x, s)
    got_hi = xs.data[:, 0].permute(0, 1, 4, 2, 3).reshape(2, 24, 9, 13)
    got_lo = xs.data[:, 1].permute(0, 1, 4, 2, 3).reshape(2, 24, 9, 13)
    assert torch.equal(got_hi, hi) and torch.equal(got_lo, lo)
    v = (x * s[:, :, None, None]).clamp(-65504, 65504)
    assert ((xs.float() - v).abs() <= v.abs() * 2.0 ** -21 + 2.0 ** -26).all()     # 22 mantissa bits


@pytest.mark.parametrize('i,o,h,w,tr,batch', [(128, 128, 64, 64, False, 1), (64, 128, 40, 72, False, 2), (256, 256, 32, 32, False, 1),
                                              (64, 64, 65, 65, True, 1), (32, 64, 128, 128, True, 1), (128, 64, 33, 47, True, 2),
                                              (128, 128, 256, 256, False, 1), (128, 64, 256, 256, True, 1),
                                              (24, 128, 256, 256, False, 1), (24, 64, 256, 256, True, 1), (512, 512, 64, 64, False, 1)])
def test_split_dma_convolution_equals_the_register_staged_form(i, o, h, w, tr, batch):
    """ia_conv2d_mfma_sx (pre-split activations, operands DMA'd into LDS) is the SAME arithmetic as ia_conv2d_mfma_s: every output
    bit is equal, for whole-tile, stream-K and fix-up tiles, with the fused epilogue, and for the split second output."""
    g = torch.Generator(device='cuda').manual_seed(11 + i + h)
    x = torch.randn(batch, i, h, w, device='cuda', generator=g) * 2
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g)
    s = torch.rand(batch, i, device='cuda', generator=g) + 0.5
    sn = torch.rand(batch, o, device='cuda', generator=g) + 0.5
    wk = hipops.pack_conv_weight_split(wt)
    d = hipops.modconv_demod(s, hipops.weight_sq_sum(wt))
    xs = hipops.act_split(x, s)
    if tr:
        # (the transposed DMA form runs on its own, larger tile: the K range of a tile is cut between stream-K workers at other
        # places, so the fp32 sums agree to summation-order level instead of bit for bit)
        want = hipops.conv2d_mfma(x, wk, styles=s, demod=d, ksize=3, transposed=True)
        got = hipops.conv2d_mfma_sx(xs, wk, demod=d, transposed=True)
        assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item()
        ref = torch.nn.functional.conv_transpose2d((x * s[:, :, None, None]).double(), wt.double().transpose(0, 1), stride=2) * d.double()[:, :, None, None]
        assert (got.double() - ref).abs().max().item() <= (want.double() - ref).abs().max().item() * 1.5 + 1e-7 * ref.abs().max().item()
        return
    bias = torch.randn(o, device='cuda', generator=g)
    noise = torch.randn(h * w, device='cuda', generator=g)
    ns = torch.full((1,), 0.3, device='cuda')
    kw = dict(demod=d, noise=noise, noise_strength=ns, bias=bias, act='lrelu', gain=1.3, clamp=4.0)
    want = hipops.conv2d_mfma(x, wk, styles=s, ksize=3, **kw)
    got, got_s = hipops.conv2d_mfma_sx(xs, wk, styles_next=sn, **kw)
    # (whole-tile launches of the DMA form pair the odd tap of a chunk with the next chunk's instead of an all-zero tap: the same
    # products, added in another order -- equal to the register-staged form to summation-order level, and as close to fp64)
    scale = hipops.conv2d_mfma(x, wk, styles=s, ksize=3, **dict(kw, clamp=None)).abs().max().item()      # (magnitude of the sums before the clamp)
    assert (got - want).abs().max().item() <= 2e-6 * scale
    hi, lo = _split_reference(got, sn)
    assert torch.equal(got_s.data[:, 0].permute(0, 1, 4, 2, 3).reshape(want.shape), hi)
    assert torch.equal(got_s.data[:, 1].permute(0, 1, 4, 2, 3).reshape(want.shape), lo)
    only_s = hipops.conv2d_mfma_sx(xs, wk, styles_next=sn, want_f32=False, **kw)
    assert torch.equal(only_s.data, got_s.data)


@pytest.mark.parametrize('b,i,o,h,w,prelu,residual', [(1, 512, 512, 8, 8, False, False), (1, 1024, 1024, 16, 16, True, True), (3, 24, 136, 12, 10, False, True),
                                                       (2, 72, 200, 9, 14, True, False), (1, 8, 128, 16, 16, False, False), (1, 512, 512, 16, 16, False, False)])
def test_low_resolution_layers_split_k_inside_the_workgroup(b, i, o, h, w, prelu, residual):
    """r06 (csrc/conv_small.h): 3x3 layers of at most 256 points -- the K range dealt to the eight waves of a workgroup, partial tiles summed
    in LDS, no slabs and no fix-up launch -- against the fp64 convolution of the operands the kernel sees: ragged point / channel counts
    (last fragments partly outside), fewer channel octets than waves, per-channel PReLU slopes, clamp, residual, both outputs; and the same
    bits on every launch."""
    assert hipops.conv_sx_supported(i, o, h, w, 3, False)
    g = torch.Generator(device='cuda').manual_seed(23 + i + h)
    x = torch.randn(b, i, h, w, device='cuda', generator=g) * 2
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g)
    s = torch.rand(b, i, device='cuda', generator=g) + 0.5
    sn = torch.rand(b, o, device='cuda', generator=g) + 0.5
    d = hipops.modconv_demod(s, hipops.weight_sq_sum(wt))
    xs = hipops.act_split(x, s)
    wk = hipops.pack_conv_weight_split(wt)
    bias = torch.randn(o, device='cuda', generator=g)
    noise = torch.randn(h * w, device='cuda', generator=g)
    ns = torch.full((1,), 0.3, device='cuda')
    slopes = (torch.rand(o, device='cuda', generator=g) * 0.5) if prelu else None
    res = torch.randn(b, o, h, w, device='cuda', generator=g) if residual else None
    hipops.PROFILE = []
    try:
        got, got_s = hipops.conv2d_mfma_sx(xs, wk, demod=d, noise=noise, noise_strength=ns, bias=bias, residual=res, act='lrelu', gain=1.3, clamp=6.0,
                                           styles_next=sn, prelu=slopes)
        again, _ = hipops.conv2d_mfma_sx(xs, wk, demod=d, noise=noise, noise_strength=ns, bias=bias, residual=res, act='lrelu', gain=1.3, clamp=6.0,
                                         styles_next=sn, prelu=slopes)
    finally:
        hipops.PROFILE = None
    ref = torch.nn.functional.conv2d(xs.float().double(), wt.double(), padding=1) * d.double()[:, :, None, None] + (noise.double() * 0.3).view(1, 1, h, w)
    ref = ref + bias.double()[None, :, None, None]
    slope = slopes.double()[None, :, None, None] if prelu else 0.2
    pre = (torch.where(ref > 0, ref, ref * slope) * 1.3)
    ref = pre.clamp(-6.0, 6.0) + (res.double() if residual else 0.0)
    assert got.shape == ref.shape and (got.double() - ref).abs().max().item() <= 3e-6 * pre.abs().max().item()
    assert torch.equal(got, again)
    hi, lo = _split_reference(got, sn)
    assert torch.equal(got_s.data[:, 0].permute(0, 1, 4, 2, 3).reshape(got.shape), hi)
    assert torch.equal(got_s.data[:, 1].permute(0, 1, 4, 2, 3).reshape(got.shape), lo)


@pytest.mark.parametrize('b,i,o,res,tr', [(1, 512, 512, 8, False), (2, 512, 512, 16, False), (1, 256, 128, 16, False), (1, 512, 512, 8, True),
                                          (2, 512, 512, 16, True), (1, 64, 64, 16, True)])
def test_split_dma_convolution_on_the_small_layers(b, i, o, res, tr):
    """r03: the 8^2 / 16^2 layers run on the fp16-pair tiles too (wide tile cut between stream-K workers; 64 x 64 transposed tile):
    against the fp64 convolution of the operands the kernel sees, with the epilogue and the split second output (stride 1)."""
    assert hipops.conv_sx_supported(i, o, res, res, 3, tr)
    g = torch.Generator(device='cuda').manual_seed(17 + i + res)
    x = torch.randn(b, i, res, res, device='cuda', generator=g) * 2
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g)
    s = torch.rand(b, i, device='cuda', generator=g) + 0.5
    sn = torch.rand(b, o, device='cuda', generator=g) + 0.5
    d = hipops.modconv_demod(s, hipops.weight_sq_sum(wt))
    xs = hipops.act_split(x, s)
    wk = hipops.pack_conv_weight_split(wt)
    xv = xs.float().double()
    if tr:
        got = hipops.conv2d_mfma_sx(xs, wk, demod=d, transposed=True)
        ref = torch.nn.functional.conv_transpose2d(xv, wt.double().transpose(0, 1), stride=2) * d.double()[:, :, None, None]
        assert got.shape == ref.shape and (got.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
        assert torch.equal(got, hipops.conv2d_mfma_sx(xs, wk, demod=d, transposed=True))          # stream-K + fix-up: deterministic
        return
    bias = torch.randn(o, device='cuda', generator=g)
    noise = torch.randn(res * res, device='cuda', generator=g)
    ns = torch.full((1,), 0.3, device='cuda')
    got, got_s = hipops.conv2d_mfma_sx(xs, wk, demod=d, noise=noise, noise_strength=ns, bias=bias, act='lrelu', gain=1.3, styles_next=sn)
    ref = torch.nn.functional.conv2d(xv, wt.double(), padding=1) * d.double()[:, :, None, None] + (noise.double() * 0.3).view(1, 1, res, res)
    ref = torch.nn.functional.leaky_relu(ref + bias.double()[None, :, None, None], 0.2) * 1.3
    assert (got.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
    hi, lo = _split_reference(got, sn)
    assert torch.equal(got_s.data[:, 0].permute(0, 1, 4, 2, 3).reshape(got.shape), hi)
    assert torch.equal(got_s.data[:, 1].permute(0, 1, 4, 2, 3).reshape(got.shape), lo)


@pytest.mark.parametrize('b,i,o,h,w', [(1, 512, 512, 32, 32), (1, 512, 512, 64, 64), (1, 512, 512, 16, 16), (4, 256, 256, 64, 64), (4, 64, 64, 256, 256),
                                       (1, 128, 128, 256, 256), (4, 128, 128, 128, 128), (2, 64, 128, 33, 47), (1, 72, 64, 100, 36)])
def test_stride_2_convolution_on_the_stride_1_tiles(b, i, o, h, w):
    """ia_conv2d_down_sx (r05): 3x3, stride 2, padding 1 -- the stride-1 tiles with the point grid over every second pixel of the input
    window (wide tile whole / stream-K, narrow whole tiles where the window does not fit beside 128 channels of weights, odd sizes) --
    against the fp64 convolution of the operands the kernel sees, with the epilogue terms the encoders use (BatchNorm scale / shift,
    per-channel PReLU, residual) and the split second output; and against the stride-1 launch sub-sampled (the r04 route)."""
    assert hipops.conv_down_supported(b, i, o, h, w)
    g = torch.Generator(device='cuda').manual_seed(23 + i + h)
    x = torch.randn(b, i, h, w, device='cuda', generator=g) * 2
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g)
    scale = torch.rand(b, o, device='cuda', generator=g) + 0.5
    bias = torch.randn(o, device='cuda', generator=g)
    slopes = torch.rand(o, device='cuda', generator=g) * 0.5
    sn = torch.rand(b, o, device='cuda', generator=g) + 0.5
    oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    res = torch.randn(b, o, oh, ow, device='cuda', generator=g)
    xs = hipops.act_split(x)
    wk = hipops.pack_conv_weight_split(wt)
    got, got_s = hipops.conv2d_down_sx(xs, wk, demod=scale, bias=bias, residual=res, act='lrelu', prelu=slopes, gain=1.1, styles_next=sn)
    ref = torch.nn.functional.conv2d(xs.float().double(), wt.double(), stride=2, padding=1) * scale.double()[:, :, None, None] + bias.double()[None, :, None, None]
    ref = torch.where(ref > 0, ref, ref * slopes.double()[None, :, None, None]) * 1.1 + res.double()
    assert got.shape == ref.shape == (b, o, oh, ow)
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    print(f'stride-2 {i}->{o} @{h}x{w} B{b}: {err:.2e} of max |ref| (vs fp64)')
    assert err <= 3e-6
    hi, lo = _split_reference(got, sn)
    assert torch.equal(got_s.data[:, 0].permute(0, 1, 4, 2, 3).reshape(got.shape), hi)
    assert torch.equal(got_s.data[:, 1].permute(0, 1, 4, 2, 3).reshape(got.shape), lo)
    assert torch.equal(got, hipops.conv2d_down_sx(xs, wk, demod=scale, bias=bias, residual=res, act='lrelu', prelu=slopes, gain=1.1))   # deterministic
    plain = hipops.conv2d_down_sx(xs, wk)
    if hipops.conv_sx_supported(i, o, h, w, 3, False) or (o >= 64 and h * w >= 1024):
        full = hipops.conv2d_mfma_sx(xs, wk)[:, :, ::2, ::2]
        assert (plain - full).abs().max().item() <= 3e-6 * full.abs().max().item()


@pytest.mark.parametrize('b,i,o,n,act', [(1, 512, 512, 8, 'lrelu'), (1, 512, 512, 4, 'lrelu'), (1, 512, 512, 2, 'linear'), (3, 70, 10, 8, 'linear'), (2, 8, 5, 2, 'lrelu')])
def test_tiny_stride_2_convolution(b, i, o, n, act):
    """ia_conv3x3_s2_tiny (the 8^2 -> 4^2 -> 2^2 -> 1^2 layers of the e4e style heads: one wave per output channel, fp32 FMAs) against the
    fp64 convolution; bias + leaky ReLU in the epilogue; deterministic; shapes it does not take are refused."""
    g = torch.Generator(device='cuda').manual_seed(31 + i + n)
    x = torch.randn(b, i, n, n, device='cuda', generator=g)
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g) / (3 * i ** 0.5)
    bias = torch.randn(o, device='cuda', generator=g)
    assert hipops.conv_tiny_supported(i, o, n, n)
    got = hipops.conv3x3_s2_tiny(x, wt, bias=bias, act=act, alpha=0.01)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), stride=2, padding=1)
    if act == 'lrelu':
        ref = torch.nn.functional.leaky_relu(ref, 0.01)
    assert got.shape == ref.shape == (b, o, n // 2, n // 2)
    err = (got.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1.0)
    print(f'tiny stride-2 {i}->{o} @{n}x{n} B{b}: {err:.2e}')
    assert err <= 2e-6
    assert torch.equal(got, hipops.conv3x3_s2_tiny(x, wt, bias=bias, act=act, alpha=0.01))
    assert not hipops.conv_tiny_supported(i, o, 16, 16) and not hipops.conv_tiny_supported(i, o, 8, 4) and not hipops.conv_tiny_supported(1024, o, 8, 8)
    with pytest.raises(RuntimeError):
        hipops.conv3x3_s2_tiny(torch.randn(1, i, 6, 6, device='cuda'), wt)


@pytest.mark.parametrize('b,c,h,w,track', [(1, 64, 128, 128, True), (4, 72, 32, 36, True), (3, 16, 5, 7, True), (2, 8, 16, 16, False), (1, 512, 16, 16, True)])
def test_train_mode_batch_norm_into_the_split_format(b, c, h, w, track):
    """ia_bn_train_split (batch statistics -> affine map -> hi / lo split in two launches, running statistics moved as
    torch.nn.BatchNorm2d moves them) against the module in fp64 and against ia_act_split of the fp64 affine map."""
    torch.manual_seed(c + h)
    bn = torch.nn.BatchNorm2d(c, track_running_stats=track).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3)
        if track:
            bn.running_mean.normal_(0, 0.5); bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(b, c, h, w) * 1.7 + 0.4
    ref = __import__('copy').deepcopy(bn).double()
    want = ref(x.double())
    bn = bn.cuda()
    xs = hipops.bn_train_split(x.cuda(), bn.weight.detach(), bn.bias.detach(), bn.running_mean if track else None, bn.running_var if track else None,
                               bn.num_batches_tracked if track else None, bn.eps, bn.momentum)
    got = xs.float().cpu().double()
    err = (got - want).abs().max().item() / max(want.abs().max().item(), 1.0)
    print(f'bn_train_split B{b} C{c} {h}x{w}: {err:.2e} of max |ref|')
    assert err <= 2e-6
    if track:
        assert (bn.running_mean.cpu().double() - ref.running_mean).abs().max().item() <= 1e-6
        assert (bn.running_var.cpu().double() - ref.running_var).abs().max().item() <= 1e-6
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    one = hipops.bn_train_split(x.cuda(), None, None, None, None, None, bn.eps, 0.1, planes=1)       # no affine parameters, one fp16 plane
    plain = torch.nn.functional.batch_norm(x.double(), None, None, training=True, eps=bn.eps)
    assert (one.float().cpu().double() - plain).abs().max().item() <= 2e-3 * max(plain.abs().max().item(), 1.0)      # (fp16 rounding)


def test_stride_2_convolution_refuses_what_it_does_not_cover():
    assert not hipops.conv_down_supported(1, 64, 32, 64, 64)        # fewer than 64 output channels: no 8-wave tile
    assert not hipops.conv_down_supported(1, 512, 512, 8, 8)        # 4^2 outputs
    assert not hipops.conv_down_supported(1, 60, 64, 64, 64)        # channel octets
    xs = hipops.act_split(torch.randn(1, 64, 64, 64, device='cuda'))
    with pytest.raises(RuntimeError):
        hipops.conv2d_down_sx(xs, hipops.pack_conv_weight_split(torch.randn(32, 64, 3, 3, device='cuda')))


@pytest.mark.parametrize('b,i,o,res,planes', [(1, 32, 256, 128, 2), (2, 16, 256, 64, 2), (1, 8, 128, 128, 2), (1, 32, 256, 128, 1)])
def test_composed_upfir_layer_equals_the_two_launch_route(b, i, o, res, planes):
    """ia_upconv2d_fir_sx (transposed convolution + resample FIR + noise + bias + lrelu as ONE stride-1 launch on the composed weight,
    depth-to-space store) against the two launches it replaces (ia_conv2d_mfma_sx transposed + ia_fir_tail_split) and against fp64."""
    from invertavatar_amd.torch_utils.ops import upfirdn2d
    g = torch.Generator(device='cuda').manual_seed(3 + i + res)
    x = torch.randn(b, i, res, res, device='cuda', generator=g) * 2
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g)
    s = torch.rand(b, i, device='cuda', generator=g) + 0.5
    sn = torch.rand(b, o, device='cuda', generator=g) + 0.5
    bias = torch.randn(o, device='cuda', generator=g)
    noise = torch.randn(4 * res * res, device='cuda', generator=g)
    ns = torch.full((1,), 0.3, device='cuda')
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
    d = hipops.modconv_demod(s, hipops.weight_sq_sum(wt))
    xs = hipops.act_split(x, s, planes=planes)
    assert hipops.upconv_fir_supported(b, i, o, res, res)
    pack = hipops.pack_conv_weight_split if planes == 2 else hipops.pack_conv_weight_h
    t = hipops.conv2d_mfma_sx(xs, pack(wt), demod=d, transposed=True)
    kw = dict(act='lrelu', clamp=None)
    want, want_s = hipops.fir_tail_split(t, f, noise, ns, bias, styles_next=sn, out_hw=(2 * res, 2 * res), pad0=(1, 1), fir_gain=4.0,
                                         act_gain=2 ** 0.5, want_f32=True, planes=planes, **kw)
    got, got_s = hipops.upconv_fir_sx(xs, pack(hipops.compose_upfir_weight(wt, f)), d, noise, ns, bias, styles_next=sn, gain=2 ** 0.5,
                                      want_f32=True, split_planes=planes, **kw)
    scale = want.abs().max().item()
    tol = 3e-6 if planes == 2 else 2e-3          # (one fp16 plane: the composed weight is rounded to fp16 once instead of the factors)
    assert got.shape == want.shape and (got - want).abs().max().item() <= tol * scale, ((got - want).abs().max().item(), scale)
    assert got_s.planes == planes and (got_s.float() - want_s.float()).abs().max().item() <= max(tol, 2e-3 if planes == 1 else 0) * scale * 1.5
    if planes == 2:
        xv = (xs.float()).double()               # the operands the kernels see
        tt = torch.nn.functional.conv_transpose2d(xv, wt.double().transpose(0, 1), stride=2) * d.double()[:, :, None, None]
        fir = torch.nn.functional.conv2d(torch.nn.functional.pad(tt, [1, 1, 1, 1]).reshape(b * o, 1, 2 * res + 3, 2 * res + 3),
                                         (4 * f.double().flip([0, 1]))[None, None]).reshape(b, o, 2 * res, 2 * res)
        ref = fir + (noise.double() * 0.3).view(1, 1, 2 * res, 2 * res)
        ref = torch.nn.functional.leaky_relu(ref + bias.double()[None, :, None, None], 0.2) * 2 ** 0.5
        assert (got.double() - ref).abs().max().item() <= 3e-6 * scale
    clamped = hipops.upconv_fir_sx(xs, pack(hipops.compose_upfir_weight(wt, f)), d, noise, ns, bias, styles_next=None, act='lrelu', gain=2 ** 0.5,
                                   clamp=0.5 * scale, want_f32=True, split_planes=planes)[0]
    assert torch.equal(clamped, got.clamp(-0.5 * scale, 0.5 * scale))


@pytest.mark.parametrize('c,res,batch', [(16, 32, 2), (32, 64, 1), (8, 256, 1), (128, 128, 1)])
def test_fir_tail_split_equals_fir_tail_then_split(c, res, batch):
    """ia_fir_tail_split = ia_upfirdn2d_bias_act (same sums, bit for bit) followed by the split of (result * styles_next)."""
    from invertavatar_amd.torch_utils.ops import upfirdn2d
    g = torch.Generator(device='cuda').manual_seed(c + res)
    t = torch.randn(batch, c, 2 * (res // 2) + 1, 2 * (res // 2) + 1, device='cuda', generator=g)
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
    noise = torch.randn(res * res, device='cuda', generator=g)
    ns = torch.full((1,), 0.25, device='cuda')
    bias = torch.randn(c, device='cuda', generator=g)
    sn = torch.rand(batch, c, device='cuda', generator=g) + 0.5
    kw = dict(out_hw=(res, res), pad0=(1, 1), fir_gain=4.0, act='lrelu', act_gain=2 ** 0.5, clamp=3.0)
    want = hipops.upfirdn2d_bias_act(t, f, noise, ns, bias, up=1, **kw)
    y, ys = hipops.fir_tail_split(t, f, noise, ns, bias, styles_next=sn, want_f32=True, **kw)
    assert torch.equal(y, want)
    hi, lo = _split_reference(want, sn)
    assert torch.equal(ys.data[:, 0].permute(0, 1, 4, 2, 3).reshape(want.shape), hi)
    assert torch.equal(ys.data[:, 1].permute(0, 1, 4, 2, 3).reshape(want.shape), lo)
    only = hipops.fir_tail_split(t, f, noise, ns, bias, styles_next=sn, **kw)
    assert torch.equal(only.data, ys.data)


def test_cond_blend_split_equals_cond_blend_then_split():
    g = torch.Generator(device='cuda').manual_seed(8)
    cond = torch.randn(2, 33, 64, 64, device='cuda', generator=g)
    cond[:, -1] = torch.rand(2, 64, 64, device='cuda', generator=g)
    x = torch.randn(2, 32, 64, 64, device='cuda', generator=g)
    sn = torch.rand(2, 32, device='cuda', generator=g) + 0.5
    want = hipops.cond_blend(cond, x)
    got = hipops.cond_blend_split(cond, x, sn, None)
    hi, lo = _split_reference(want, sn)
    assert torch.equal(got.data[:, 0].permute(0, 1, 4, 2, 3).reshape(want.shape), hi)
    assert torch.equal(got.data[:, 1].permute(0, 1, 4, 2, 3).reshape(want.shape), lo)



@pytest.mark.parametrize('i,o,h,w,tr', [(128, 128, 64, 64, False), (256, 128, 48, 80, False), (64, 64, 65, 65, True), (32, 256, 128, 128, True)])
def test_one_plane_dma_convolution_equals_the_fp16_operand_form(i, o, h, w, tr):
    """The fp16-storage form (one fp16 plane in, one product per k-step) is the arithmetic of ia_conv2d_mfma_h (same products, equal to
    summation-order level), with the activation stored as 2 bytes per element; its split second output is the rounded fp16 plane of
    (result * styles_next)."""
    g = torch.Generator(device='cuda').manual_seed(5 + i + h)
    x = torch.randn(2, i, h, w, device='cuda', generator=g) * 2
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g)
    s = torch.rand(2, i, device='cuda', generator=g) + 0.5
    sn = torch.rand(2, o, device='cuda', generator=g) + 0.5
    wk = hipops.pack_conv_weight_h(wt)
    d = hipops.modconv_demod(s, hipops.weight_sq_sum(wt))
    xs = hipops.act_split(x, s, planes=1)
    assert xs.data.shape[1] == 1 and torch.equal(xs.data[:, 0].permute(0, 1, 4, 2, 3).reshape(x.shape), (x * s[:, :, None, None]).half())
    if tr:
        want = hipops.conv2d_mfma(x, wk, styles=s, demod=d, ksize=3, transposed=True)
        got = hipops.conv2d_mfma_sx(xs, wk, demod=d, transposed=True)
        assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item()
        return
    bias = torch.randn(o, device='cuda', generator=g)
    kw = dict(demod=d, bias=bias, act='lrelu', gain=1.3, clamp=256.0)
    want = hipops.conv2d_mfma(x, wk, styles=s, ksize=3, **kw)
    got, got_s = hipops.conv2d_mfma_sx(xs, wk, styles_next=sn, split_planes=1, **kw)
    # (the DMA form pairs the odd tap of a chunk with the next chunk's: the same products in another order)
    assert (got - want).abs().max().item() <= 2e-6 * want.abs().max().item()
    assert got_s.planes == 1 and torch.equal(got_s.data[:, 0].permute(0, 1, 4, 2, 3).reshape(want.shape), (got * sn[:, :, None, None]).half())


@pytest.mark.parametrize('b,i,o,h,w', [(1, 512, 32, 64, 64), (1, 256, 96, 128, 128), (2, 128, 3, 256, 256), (1, 128, 96, 256, 256),
                                       (1, 64, 40, 36, 36), (3, 32, 32, 6, 6), (1, 128, 3, 512, 512)])
def test_conv1x1_streaming_matches_torch(b, i, o, h, w):
    """ia_conv1x1 (ToRGB: modulated 1x1 convolution, bias, clamp, skip add) against the reference's op order in torch fp64
    and against the tiled ia_conv2d_mfma form it replaces (same products, different summation order)."""
    from conftest import rnd
    x, wgt = rnd(1, b, i, h, w).cuda(), (rnd(2, o, i, 1, 1) / i ** 0.5).cuda()
    styles, bias, res = (rnd(3, b, i) * 0.3 + 1).cuda(), rnd(4, o).cuda(), rnd(5, b, o, h, w).cuda()
    wk = hipops.pack_conv_weight(wgt)
    clamp = 1.5
    ref = torch.einsum('bihw,boi->bohw', x.double(), wgt.double()[None, :, :, 0, 0] * styles.double()[:, None, :])
    ref = (ref + bias.double()[None, :, None, None]).clamp(-clamp, clamp) + res.double()
    got = hipops.conv1x1(x, wk, styles, bias=bias, residual=res, clamp=clamp)
    assert got.shape == ref.shape and max_abs(got.double(), ref) <= 2e-5, max_abs(got.double(), ref)
    tiled = hipops.conv2d_mfma(x, wk, styles, None, bias=bias, residual=res, ksize=1, act='linear', clamp=clamp)
    assert max_abs(got, tiled) <= 2e-5
    plain = hipops.conv1x1(x, wk)                                   # no styles / bias / residual / clamp
    assert max_abs(plain.double(), torch.einsum('bihw,oi->bohw', x.double(), wgt.double()[:, :, 0, 0])) <= 2e-5


@pytest.mark.parametrize('b,i,o,h,w', [(1, 512, 32, 4, 4), (1, 512, 96, 8, 8), (2, 512, 32, 16, 16), (1, 512, 96, 32, 32), (1, 512, 32, 64, 64),
                                       (1, 256, 96, 128, 128), (2, 128, 32, 256, 256), (1, 128, 3, 512, 512), (1, 1024, 8, 6, 10), (1, 256, 5, 2, 2), (1, 256, 3, 256, 256), (2, 512, 4, 128, 128),
                                       (1, 128, 96, 256, 256), (1, 128, 35, 128, 128), (2, 256, 64, 128, 128), (1, 128, 20, 136, 124), (1, 256, 32, 128, 128)])
def test_torgb_with_fused_skip_upsampling(b, i, o, h, w):
    """ia_torgb (ToRGB + upsample2d(previous image) + add in one launch) against the reference's op order in torch fp64, against the
    two-launch route it replaces (ia_conv1x1 + ia_upfirdn2d: the same up-sampled image bit for bit), and with a plain residual."""
    from conftest import rnd
    from invertavatar_amd.torch_utils.ops import upfirdn2d
    x, wgt = rnd(1, b, i, h, w).cuda(), (rnd(2, o, i, 1, 1) / i ** 0.5).cuda()
    styles, bias, skip = (rnd(3, b, i) * 0.3 + 1).cuda(), rnd(4, o).cuda(), rnd(5, b, o, h // 2, w // 2).cuda()
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
    wk = hipops.pack_conv_weight(wgt)
    clamp = 1.5
    assert hipops.torgb_supported(i, o, h, w, True)
    up = upfirdn2d.upsample2d(skip, f)
    conv = torch.einsum('bihw,boi->bohw', x.double(), wgt.double()[None, :, :, 0, 0] * styles.double()[:, None, :])
    conv = (conv + bias.double()[None, :, None, None]).clamp(-clamp, clamp)
    got = hipops.torgb(x, wk, styles, bias=bias, skip=skip, skip_filter=f, clamp=clamp)
    assert got.shape == conv.shape and max_abs(got.double(), conv + up.double()) <= 2e-5, max_abs(got.double(), conv + up.double())
    # the up-sampled image alone (zero weights): bit-identical to ia_upfirdn2d's
    only_up = hipops.torgb(x, torch.zeros_like(wk), None, skip=skip, skip_filter=f)
    assert torch.equal(only_up, up)
    res = rnd(6, b, o, h, w).cuda()
    got_r = hipops.torgb(x, wk, styles, bias=bias, residual=res, clamp=clamp)
    assert max_abs(got_r.double(), conv + res.double()) <= 2e-5
    if hipops.conv1x1_supported(i, o, h, w):
        assert max_abs(got_r, hipops.conv1x1(x, wk, styles, bias=bias, residual=res, clamp=clamp)) <= 2e-5
    plain = hipops.torgb(x, wk)                                     # no styles / bias / skip / clamp
    assert max_abs(plain.double(), torch.einsum('bihw,oi->bohw', x.double(), wgt.double()[:, :, 0, 0])) <= 2e-5
    with pytest.raises(RuntimeError, match='ia_torgb covers'):
        hipops.torgb(torch.zeros(1, 48, 8, 8, device='cuda'), torch.zeros(1, 48, 8, device='cuda'))


@pytest.mark.parametrize('i,o,r', [(128, 96, 128), (256, 64, 128)])
def test_torgb_wide_channel_blocks_on_fp16_pairs(i, o, r):
    """ia_torgb with 64 / 96 output channels on a large image (torgb_wide_kernel, r06) forms its fp32 products from fp16 hi / lo pairs:
    activations over eight decades (tiny ones ride in the scaled low part), error against fp64 at the level of fp32 rounding; an
    activation outside the fp16 range raises the library's range-watch word and nothing else does."""
    from conftest import rnd
    mag = 10.0 ** (rnd(31, 1, i, r, r) * 2.0 - 2.0).clamp(-6, 2.5)
    x = (rnd(32, 1, i, r, r) * mag).cuda()
    wgt = (rnd(33, o, i, 1, 1) / i ** 0.5).cuda()
    styles, bias = (rnd(34, 1, i) * 0.3 + 1).cuda(), rnd(35, o).cuda()
    wk = hipops.pack_conv_weight(wgt)
    ref = torch.einsum('bihw,boi->bohw', x.double(), wgt.double()[None, :, :, 0, 0] * styles.double()[:, None, :]) + bias.double()[None, :, None, None]
    bound = torch.einsum('bihw,boi->bohw', x.double().abs(), (wgt.double()[None, :, :, 0, 0] * styles.double()[:, None, :]).abs())
    hipops.split_saturation_poll()
    got = hipops.torgb(x, wk, styles, bias=bias)
    ratio = ((got.double() - ref).abs() / (6e-7 * bound + 1e-7 * ref.abs() + 1e-9)).max().item()
    assert ratio <= 1.0, ratio
    assert not hipops.split_saturation_poll()
    x[0, 5, 7, 9] = 1e5
    hipops.torgb(x, wk, styles, bias=bias)
    assert hipops.split_saturation_poll()


def test_conv1x1_rejects_unsupported_shapes():
    x = torch.zeros(1, 48, 8, 8, device='cuda')
    with pytest.raises(RuntimeError, match='ia_conv1x1 covers'):
        hipops.conv1x1(x, torch.zeros(1, 48, 8, device='cuda'))
    assert not hipops.conv1x1_supported(48, 8, 8, 8) and not hipops.conv1x1_supported(64, 128, 8, 8) and hipops.conv1x1_supported(64, 96, 8, 8)


@pytest.mark.parametrize('res,planes', [(256, 2), (512, 2), (256, 1)])
def test_conv_sx_with_fused_torgb(res, planes):
    """ia_conv2d_mfma_sx_rgb = ia_conv2d_mfma_sx followed by ia_conv1x1 on its result (the last SR block: conv1, ToRGB, skip add)."""
    b, c, rc = 1, 128, 3
    x, styles = rnd(11, b, c, res, res).cuda(), (rnd(12, b, c) * 0.3 + 1).cuda()
    w3 = (rnd(13, c, c, 3, 3) / (9 * c) ** 0.5).cuda()
    wk = hipops.pack_conv_weight_split(w3) if planes == 2 else hipops.pack_conv_weight_h(w3)
    demod = (rnd(14, b, c).abs() + 0.5).cuda()
    noise, ns, bias = rnd(15, res * res).cuda(), torch.tensor([0.3], device='cuda'), rnd(16, c).cuda()
    rgb_w = hipops.pack_conv_weight((rnd(17, rc, c, 1, 1) / c ** 0.5).cuda())
    rgb_styles, rgb_bias, skip = (rnd(18, b, c) * 0.3 + 1).cuda(), rnd(19, rc).cuda(), rnd(20, b, rc, res, res).cuda()
    assert hipops.conv_sx_rgb_supported(b, c, c, res, res)
    xs = hipops.act_split(x, styles, planes=planes)
    kw = dict(act='lrelu', gain=2 ** 0.5, clamp=256)
    y = hipops.conv2d_mfma_sx(xs, wk, demod, noise, ns, bias, **kw)
    ref = hipops.conv1x1(y, rgb_w, rgb_styles, bias=rgb_bias, residual=skip, clamp=1.5)
    y2, ys2, got = hipops.conv2d_mfma_sx_rgb(xs, wk, rgb_w, rgb_styles, rgb_bias, skip, 1.5, demod, noise, ns, bias, **kw)
    assert y2 is None and ys2 is None and got.shape == ref.shape
    assert max_abs(got, ref) <= 2e-5, max_abs(got, ref)
    y3, _, got3 = hipops.conv2d_mfma_sx_rgb(xs, wk, rgb_w, rgb_styles, rgb_bias, skip, 1.5, demod, noise, ns, bias, want_f32=True, **kw)
    assert torch.equal(y3, y) and torch.equal(got3, got)         # the layer's own output is unchanged by the extra epilogue
    assert not hipops.conv_sx_rgb_supported(1, 128, 128, 128, 128) and not hipops.conv_sx_rgb_supported(1, 256, 256, 256, 256)
    with pytest.raises(RuntimeError, match='fused ToRGB'):
        small = hipops.act_split(x[:, :, :128, :128].contiguous(), styles, planes=planes)
        hipops.conv2d_mfma_sx_rgb(small, wk, rgb_w, rgb_styles, rgb_bias, None, 1.5, demod)


@pytest.mark.parametrize('b,i,o,h,w', [(1, 32, 128, 64, 64), (2, 48, 256, 24, 40), (1, 16, 128, 16, 256), (1, 64, 128, 130, 128),
                                       (1, 512, 256, 64, 64), (1, 256, 128, 128, 128), (3, 32, 128, 33, 17)])
def test_row_phase_upconv_equals_the_four_phase_form_and_fp64(b, i, o, h, w):
    """r04: ia_upconv2d_rows_sx -- the transposed 3x3 convolution per output row phase on the stride-1 tile (two accumulator sets, no
    zero k-step) -- forms the same products as ia_conv2d_mfma_sx(transposed): equal to summation-order level, as close to the fp64
    convolution of the operands the kernel sees, every pixel of the (2H+1) x (2W+1) image written (interior tiles, bottom row, last
    column; whole tiles and K-split edge tiles), run-to-run identical."""
    assert hipops.upconv_rows_supported(b, i, o, h, w)
    g = torch.Generator(device='cuda').manual_seed(23 + i + h)
    x = torch.randn(b, i, h, w, device='cuda', generator=g) * 2
    wt = torch.randn(o, i, 3, 3, device='cuda', generator=g)
    s = torch.rand(b, i, device='cuda', generator=g) + 0.5
    d = hipops.modconv_demod(s, hipops.weight_sq_sum(wt))
    xs = hipops.act_split(x, s)
    wk = hipops.pack_conv_weight_split(wt)
    ref = torch.nn.functional.conv_transpose2d(xs.float().double(), wt.double().transpose(0, 1), stride=2) * d.double()[:, :, None, None]
    scale = ref.abs().max().item()
    poison = torch.full_like(ref, float('nan'), dtype=torch.float32)
    got = hipops.upconv2d_rows_sx(xs, wk, demod=d)
    assert got.shape == ref.shape and torch.isfinite(got).all()
    err = (got.double() - ref).abs().max().item()
    assert err <= 3e-6 * scale, err / scale
    if hipops.conv_sx_supported(i, o, h, w, 3, True):
        four = hipops.conv2d_mfma_sx(xs, wk, demod=d, transposed=True)
        assert (got - four).abs().max().item() <= 2e-6 * scale
        assert err <= (four.double() - ref).abs().max().item() * 1.5 + 1e-7 * scale
    again = hipops.upconv2d_rows_sx(xs, wk, demod=d)
    assert torch.equal(got, again)
    none = hipops.upconv2d_rows_sx(xs, wk)                      # without demodulation
    ref1 = torch.nn.functional.conv_transpose2d(xs.float().double(), wt.double().transpose(0, 1), stride=2)
    assert (none.double() - ref1).abs().max().item() <= 3e-6 * ref1.abs().max().item()
    del poison


def test_row_phase_upconv_refuses_what_it_does_not_cover():
    assert not hipops.upconv_rows_supported(1, 24, 128, 64, 64)       # I % 16
    assert not hipops.upconv_rows_supported(1, 32, 64, 64, 64)        # O % 128
    assert not hipops.upconv_rows_supported(1, 32, 128, 8, 8)         # below 16^2
    xs = hipops.act_split(torch.randn(1, 32, 32, 32, device='cuda'))
    wk = hipops.pack_conv_weight_split(torch.randn(64, 32, 3, 3, device='cuda'))
    with pytest.raises(RuntimeError, match='row-phase'):
        hipops.upconv2d_rows_sx(xs, wk)
