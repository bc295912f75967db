"""filtered_lrelu: oracle and the product's torch route against outputs of the reference's own definition of
the op on CPU; the fused HIP kernel (ia_filtered_lrelu) against the same vectors on the GPU."""
import pytest
import torch

from oracle import ops as O
from invertavatar_amd.torch_utils.ops import filtered_lrelu as flr, _plugins
from conftest import max_abs

CASES = {   # as tests/golden/make_golden.py:FLR_CASES -> (up, down, padding, gain, slope, clamp, flip)
    'sg3_up2_down2': (2, 2, [10, 11, 9, 10], 2 ** 0.5, 0.2, 256.0, False),
    'up4_down2_flip': (4, 2, [5, 6, 7, 4], 1.7, 0.1, None, True),
    'up2_only_2d': (2, 1, [2, 1, 2, 1], 2 ** 0.5, 0.2, 0.8, False),
    'down2_only': (1, 2, 0, 1.0, 0.3, None, False),
    'identity_filters': (1, 1, [1, -1, 0, 2], 2 ** 0.5, 0.2, 1.0, False),
    'crop_negative_pad': (2, 2, [-3, 4, 2, -1], 2 ** 0.5, 0.2, None, False),
}


def _case(g, name):
    keys = g.keys()
    get = lambda k: g[f'{name}/{k}'] if f'{name}/{k}' in keys else None   # noqa: E731
    up, down, pad, gain, slope, clamp, flip = CASES[name]
    return get('x'), get('y'), dict(fu=get('fu'), fd=get('fd'), b=get('b'), up=up, down=down, padding=pad, gain=gain, slope=slope, clamp=clamp,
                                    flip_filter=flip)


@pytest.mark.parametrize('name', list(CASES))
def test_oracle_and_torch_route_match_the_reference(golden, name):
    x, y, kw = _case(golden('filtered_lrelu.npz'), name)
    assert max_abs(O.filtered_lrelu(x, **kw), y) <= 2e-6
    got = flr.filtered_lrelu(x, impl='ref', **kw)
    assert got.shape == y.shape and max_abs(got, y) <= 2e-6
    assert max_abs(flr.filtered_lrelu(x, **kw), y) <= 2e-6          # impl='cuda' on a CPU tensor takes the same route


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CASES))
def test_fused_kernel_matches_the_reference(golden, name):
    x, y, kw = _case(golden('filtered_lrelu.npz'), name)
    dev = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
    got = flr.filtered_lrelu(x.cuda(), **dev)
    assert got.shape == y.shape and max_abs(got.cpu(), y) <= 2e-6
    # the plugin-shaped entry point (pybind argument list of filtered_lrelu.cpp:20) reports "kernel ran"
    px0, px1, py0, py1 = flr._parse_padding(kw['padding'])
    out, so, rc = _plugins.filtered_lrelu(x.cuda(), dev['fu'], dev['fd'], dev['b'], None, kw['up'], kw['down'], px0, px1, py0, py1, 0, 0,
                                          kw['gain'], kw['slope'], -1 if kw['clamp'] is None else kw['clamp'], kw['flip_filter'], False)
    assert rc == 0 and torch.equal(out, got) and so.numel() == 0
    # fp16 storage: computed in fp32, rounded once
    h = flr.filtered_lrelu(x.cuda().half(), **{k: (v.half() if k == 'b' and v is not None else v) for k, v in dev.items()})
    assert h.dtype == torch.float16 and max_abs(h.float().cpu(), y) <= 2e-2 * max(1.0, y.abs().max().item())


@pytest.mark.gpu
def test_fused_kernel_on_a_large_plane_and_fallbacks():
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 8, 130, 250, generator=g)
    fu, fd, b = torch.rand(12, generator=g) / 6, torch.rand(12, generator=g) / 6, torch.randn(8, generator=g)
    kw = dict(up=2, down=2, padding=[10, 11, 10, 11], gain=2 ** 0.5, slope=0.2, clamp=256.0)
    ref = O.filtered_lrelu(x, fu, fd, b, **kw)
    got = flr.filtered_lrelu(x.cuda(), fu.cuda(), fd.cuda(), b.cuda(), **kw)
    assert max_abs(got.cpu(), ref) <= 1e-5
    # float64 and sign-tensor calls have no kernel: return code -1 / composed route, as filtered_lrelu.py:225-231
    _, _, rc = _plugins.filtered_lrelu(x.cuda().double(), fu.cuda(), fd.cuda(), b.cuda().double(), None, 2, 2, 10, 11, 10, 11, 0, 0, 1.4, 0.2, -1, False, False)
    assert rc == -1
    d = flr.filtered_lrelu(x.cuda().double(), fu.cuda(), fd.cuda(), b.cuda().double(), **kw)
    assert d.dtype == torch.float64 and max_abs(d.cpu(), ref) <= 1e-5
