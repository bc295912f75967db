"""Parity of the HIP operator kernels against the CPU oracle (through the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import ops as O
from invertavatar_amd.torch_utils.ops import bias_act, upfirdn2d, filtered_lrelu
from conftest import rnd, max_abs

pytestmark = pytest.mark.gpu
ACTS = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']


@pytest.mark.parametrize('act', ACTS)
def test_bias_act_forward_all_activations(act, golden):
    x, b = rnd(1, 2, 6, 9, 7) * 2, rnd(2, 6)
    y = bias_act.bias_act(x.cuda(), b.cuda(), act=act).cpu()
    assert max_abs(y, O.bias_act(x, b, act=act)) <= 2e-6
    assert max_abs(y, golden('ops.npz')[f'bias_act/{act}']) <= 2e-6


@pytest.mark.parametrize('shape,dim', [((1, 128, 64, 64), 1), ((3, 7, 5, 3), 1), ((4, 33), 1), ((2, 5, 6, 7), 3), ((1, 1, 1, 1), 1)])
def test_bias_act_shapes_clamp_and_layouts(shape, dim):
    x = rnd(3, *shape) * 3
    b = rnd(4, shape[dim])
    ref = O.bias_act(x, b, dim=dim, act='lrelu', gain=1.3, clamp=2.0)
    assert max_abs(bias_act.bias_act(x.cuda(), b.cuda(), dim=dim, act='lrelu', gain=1.3, clamp=2.0).cpu(), ref) <= 2e-6
    if len(shape) == 4:
        xc = x.cuda().contiguous(memory_format=torch.channels_last)
        y = bias_act.bias_act(xc, b.cuda(), dim=dim, act='lrelu', gain=1.3, clamp=2.0)
        assert y.stride() == xc.stride()
        assert max_abs(y.cpu(), ref) <= 2e-6


def test_bias_act_f16_and_f64():
    x, b = rnd(5, 2, 8, 16, 16), rnd(6, 8)
    ref = O.bias_act(x.half().float(), b.half().float(), act='lrelu', clamp=256).half()
    y = bias_act.bias_act(x.cuda().half(), b.cuda().half(), act='lrelu', clamp=256)
    assert y.dtype == torch.float16 and max_abs(y.cpu(), ref) <= 2e-3
    y64 = bias_act.bias_act(x.cuda().double(), b.cuda().double(), act='softplus')
    assert max_abs(y64.cpu(), O.bias_act(x.double(), b.double(), act='softplus')) <= 1e-12


def test_bias_act_gradients_match_autograd():
    for act in ('lrelu', 'sigmoid', 'swish', 'softplus'):
        x = (rnd(7, 2, 4, 5, 5)).cuda().requires_grad_(True)
        b = rnd(8, 4).cuda().requires_grad_(True)
        y = bias_act.bias_act(x, b, act=act, clamp=1.5)
        gx, gb = torch.autograd.grad(y.square().sum(), [x, b])
        xr, br = x.detach().cpu().requires_grad_(True), b.detach().cpu().requires_grad_(True)
        yr = bias_act.bias_act(xr, br, act=act, clamp=1.5, impl='ref')
        gxr, gbr = torch.autograd.grad(yr.square().sum(), [xr, br])
        assert max_abs(gx.cpu(), gxr) <= 1e-5 and max_abs(gb.cpu(), gbr) <= 1e-4, act


def test_bias_act_empty_and_errors():
    assert bias_act.bias_act(torch.empty(0, 4).cuda(), torch.zeros(4).cuda()).shape == (0, 4)
    with pytest.raises(RuntimeError):
        bias_act.bias_act(torch.zeros(2, 3, 4, 4).cuda(), torch.zeros(5).cuda())   # wrong bias length


def test_upfirdn2d_golden_cases(golden):
    g = golden('ops.npz')
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
    x = rnd(4, 2, 5, 13, 11).cuda()
    assert max_abs(upfirdn2d.upfirdn2d(x, f, padding=[1, 1, 1, 1], gain=4).cpu(), g['upfirdn2d/blur_pad1']) <= 2e-6
    assert max_abs(upfirdn2d.upsample2d(x, f).cpu(), g['upfirdn2d/up2']) <= 2e-6
    assert max_abs(upfirdn2d.downsample2d(x, f).cpu(), g['upfirdn2d/down2']) <= 2e-6
    y = upfirdn2d.upfirdn2d(x, rnd(5, 3, 5).abs().cuda(), up=[2, 3], down=[3, 2], padding=[2, -1, 0, 3], flip_filter=True, gain=1.5)
    assert max_abs(y.cpu(), g['upfirdn2d/mixed']) <= 5e-6
    y = upfirdn2d.upfirdn2d(x, (torch.tensor([1., 3., 3., 1.]) / 8).cuda(), up=2, padding=[2, 1, 2, 1], gain=4)
    assert max_abs(y.cpu(), g['upfirdn2d/sep']) <= 2e-6


@pytest.mark.parametrize('n,c,h,w', [(1, 3, 128, 128), (2, 5, 33, 65), (1, 2, 4, 4), (1, 1, 1, 1), (1, 96, 64, 64)])
def test_upfirdn2d_hot_shapes(n, c, h, w):
    """The two specialisations the generator hits: post-transposed-conv blur and 2x skip up-sampling."""
    f = O.setup_filter([1, 3, 3, 1])
    x = rnd(9, n, c, 2 * h + 1, 2 * w + 1)
    y = upfirdn2d.upfirdn2d(x.cuda(), f.cuda(), padding=[1, 1, 1, 1], gain=4)
    assert y.shape == (n, c, 2 * h, 2 * w) and max_abs(y.cpu(), O.upfirdn2d(x, f, padding=(1, 1, 1, 1), gain=4)) <= 3e-6
    x = rnd(10, n, c, h, w)
    y = upfirdn2d.upsample2d(x.cuda(), f.cuda())
    assert y.shape == (n, c, 2 * h, 2 * w) and max_abs(y.cpu(), O.upsample2d(x, f)) <= 3e-6


def test_upfirdn2d_channels_last_f16_and_linearity():
    f = O.setup_filter([1, 3, 3, 1])
    x = rnd(11, 2, 8, 20, 24)
    ref = O.upsample2d(x, f)
    xc = x.cuda().contiguous(memory_format=torch.channels_last)
    y = upfirdn2d.upsample2d(xc, f.cuda())
    assert y.is_contiguous(memory_format=torch.channels_last) and max_abs(y.cpu(), ref) <= 3e-6
    yh = upfirdn2d.upsample2d(x.cuda().half(), f.cuda())
    assert yh.dtype == torch.float16 and max_abs(yh.float().cpu(), O.upsample2d(x.half().float(), f)) <= 4e-3
    # size-independent property at the generator's largest shape: linearity
    a, b = torch.randn(1, 16, 513, 513, device='cuda'), torch.randn(1, 16, 513, 513, device='cuda')
    op = lambda t: upfirdn2d.upfirdn2d(t, f.cuda(), padding=[1, 1, 1, 1], gain=4)
    assert (op(a + 2 * b) - (op(a) + 2 * op(b))).abs().max().item() <= 2e-5
    # DC gain: a constant image stays constant away from the border (filter sums to 1, gain 4 / up 2^2)
    ones = upfirdn2d.upsample2d(torch.ones(1, 1, 64, 64, device='cuda'), f.cuda())
    assert (ones[:, :, 2:-2, 2:-2] - 1).abs().max().item() <= 1e-6


def test_filtered_lrelu_composition_on_device():
    x, b = rnd(12, 1, 4, 16, 16), rnd(13, 4)
    fu = O.setup_filter([1, 3, 3, 1])
    y = filtered_lrelu.filtered_lrelu(x.cuda(), fu.cuda(), fu.cuda(), b.cuda(), up=2, down=2, padding=3)
    r = filtered_lrelu.filtered_lrelu(x, fu, fu, b, up=2, down=2, padding=3, impl='ref')
    assert max_abs(y.cpu(), r) <= 1e-5


def test_cond_blend_is_the_reference_expression_bit_for_bit():
    """ia_cond_blend = cond[:, :-1] * a + x * (1 - a), a = cond[:, -1:] (networks_stylegan2_new.py:537-540): same
    operation order as the four elementwise kernels of the reference, so the bits must be identical."""
    from invertavatar_amd import hipops
    cond, x = rnd(1, 2, 33, 16, 24).cuda(), rnd(2, 2, 32, 16, 24).cuda()
    cond[:, -1:] = cond[:, -1:].sigmoid()
    a = cond[:, -1:]
    ref = cond[:, :-1] * a + x * (1 - a)
    assert torch.equal(hipops.cond_blend(cond, x), ref)


def test_stage_inputs_copies_segments_in_one_launch():
    from invertavatar_amd import hipops
    """ia_stage_inputs: aligned and unaligned segments, odd byte counts, a strided source (falls through to copy_)."""
    torch.manual_seed(3)
    big = torch.randn(128 * 128 * 48 + 3, device='cuda')
    srcs = [torch.randn(1, 14, 512, device='cuda'), torch.randn(1, 25, device='cuda'), big[1:],        # big[1:]: 4-byte aligned only
            torch.randint(0, 255, (4099,), device='cuda', dtype=torch.uint8), torch.randn(6, 10, device='cuda')[:, ::2]]
    dsts = [torch.zeros_like(s, memory_format=torch.contiguous_format) for s in srcs]
    hipops.stage_inputs(list(zip(srcs, dsts)))
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(s, d)


@pytest.mark.parametrize('shape', [(1, 32, 32, 32), (2, 512, 64, 64), (1, 256, 128, 128), (1, 40, 7, 9)])
def test_channels_last_copy_kernel(shape):
    """ia_channels_last = permute(0, 2, 3, 1).contiguous(), bit for bit (ragged channel / pixel counts included)."""
    from invertavatar_amd import hipops
    x = torch.randn(*shape, device='cuda')
    got = hipops.channels_last_copy(x)
    assert got.is_contiguous() and torch.equal(got, x.permute(0, 2, 3, 1).contiguous())
