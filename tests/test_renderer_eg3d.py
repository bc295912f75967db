"""`ImportanceRenderer` (reference renderer.py:122-293: per-ray box limits, flip_z, stochastic importance draws) and
`math_utils.get_ray_limits_box` / `linspace` (math_utils.py:46-118) against `renderer_eg3d.npz`, which tests/golden/make_golden.py
records from the reference class itself with both random draws pinned.  CPU: the oracle restatement and the product class;
`-m gpu`: the product class on device tensors."""
import numpy as np
import pytest
import torch

from oracle import renderer as OR
from invertavatar_amd import synthetic
from invertavatar_amd.training_avatar_texture.triplane_v20 import OSGDecoder
from invertavatar_amd.training_avatar_texture.volumetric_rendering import math_utils
from invertavatar_amd.training_avatar_texture.volumetric_rendering.renderer import ImportanceRenderer
from conftest import rnd, max_abs, fixed_randomness

CASES = {'auto': (False, 'auto', 'auto'), 'auto_flip': (True, 'auto', 'auto'), 'fixed': (False, 2.25, 3.3), 'fixed_flip': (True, 2.25, 3.3)}
N_FINE = 48


def _inputs(g):
    frames, nrr = g['frames'].tolist(), g['nrr']
    return rnd(21, 2, 3, 32, 48, 48), synthetic.jitter(frames, nrr * nrr), g['rays_o'], g['rays_d']


def _decoder_state():
    sd = {k: torch.empty(s) for k, s in (('net.0.weight', (64, 32)), ('net.0.bias', (64,)), ('net.2.weight', (33, 64)), ('net.2.bias', (33,)))}
    return synthetic.fill_parameters(sd, salt=5)


def _decoder_module():
    dec = OSGDecoder(32, {'decoder_lr_mul': 1, 'decoder_output_dim': 32}).eval().requires_grad_(False)
    return synthetic.fill_parameters(dec, salt=5)


def _uniform(n_rays):
    return torch.from_numpy(np.random.RandomState(99).rand(n_rays, N_FINE).astype(np.float32))


def test_ray_limits_box_and_linspace(golden):
    g = golden('renderer_eg3d.npz')
    for fn in (OR.ray_limits_box, math_utils.get_ray_limits_box):
        t0, t1 = fn(g['rays_o'], g['rays_d'], 1)
        assert torch.equal(t0, g['box_near']) and torch.equal(t1, g['box_far'])
    miss = g['box_far'] <= g['box_near']
    assert miss.any() and not miss.all() and (g['box_near'][miss] == -1).all() and (g['box_far'][miss] == -2).all()
    # axis-parallel rays (1/d = +-inf on two axes), from inside and outside the slabs
    o = torch.tensor([[[0.1, -0.2, 2.0], [0.7, 0.0, 2.0], [0.0, 0.0, -3.0], [0.2, 0.3, 0.1]]])
    d = torch.tensor([[[0.0, 0.0, -1.0], [0.0, 0.0, -1.0], [-0.0, 0.0, 1.0], [1.0, 0.0, -0.0]]])
    a0, a1 = OR.ray_limits_box(o, d, 1)
    b0, b1 = math_utils.get_ray_limits_box(o, d, 1)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
    assert a0[0, :, 0].tolist() == [1.5, -1.0, 2.5, -0.699999988079071] and a1[0, :, 0].tolist() == [2.5, -2.0, 3.5, 0.30000001192092896]
    s, e = torch.tensor([[0.5], [2.0]]), torch.tensor([[1.5], [-1.0]])
    lin = math_utils.linspace(s, e, 5)
    assert lin.shape == (5, 2, 1) and torch.equal(lin[:, 0, 0], torch.tensor([0.5, 0.75, 1.0, 1.25, 1.5])) and lin[-1, 1, 0] == -1.0


@pytest.mark.parametrize('case', list(CASES))
def test_oracle_eg3d_renderer_vs_reference(golden, case):
    g = golden('renderer_eg3d.npz')
    planes, jit, ro, rd = _inputs(g)
    flip, start, end = CASES[case]
    rgb, depth, wsum, aux = OR.render_eg3d(planes, _decoder_state(), ro, rd, jit, _uniform(ro.shape[0] * ro.shape[1]), start, end, flip,
                                           return_aux=True)
    assert torch.equal(aux['z_coarse'], g[f'{case}/z_coarse'])
    assert max_abs(aux['w_coarse'], g[f'{case}/w_coarse']) <= 1e-5
    assert max_abs(rgb, g[f'{case}/rgb']) <= 1e-5 and max_abs(depth, g[f'{case}/depth']) <= 1e-5 and max_abs(wsum, g[f'{case}/wsum']) <= 1e-5
    # integer buffers: identical given the reference's own stage inputs (stage level, as for R7 / R8)
    _, ibuf = OR.sample_importance(g[f'{case}/z_coarse'], g[f'{case}/w_coarse'], N_FINE, u=g[f'{case}/u'])
    assert torch.equal(ibuf['inds'], g[f'{case}/inds'])
    assert torch.equal(g[f'{case}/u'], _uniform(g[f'{case}/u'].shape[0]))


def _run_product(g, case, device):
    planes, jit, ro, rd = _inputs(g)
    flip, start, end = CASES[case]
    ren = ImportanceRenderer(flip_z=flip)
    dec = _decoder_module().to(device)
    with torch.no_grad(), fixed_randomness(jit):
        return ren(planes.to(device), dec, ro.to(device).clone(), rd.to(device).clone(), synthetic.rendering_kwargs(ray_start=start, ray_end=end))


@pytest.mark.parametrize('case', list(CASES))
def test_product_eg3d_renderer_on_cpu_vs_reference(golden, case):
    g = golden('renderer_eg3d.npz')
    rgb, depth, wsum = _run_product(g, case, 'cpu')
    assert rgb.shape == g[f'{case}/rgb'].shape and depth.shape == g[f'{case}/depth'].shape and wsum.shape == g[f'{case}/wsum'].shape
    assert max_abs(rgb, g[f'{case}/rgb']) <= 1e-5 and max_abs(depth, g[f'{case}/depth']) <= 1e-5 and max_abs(wsum, g[f'{case}/wsum']) <= 1e-5


def test_flip_z_and_auto_limits_change_the_image(golden):
    """The four cases are four different images: a renderer that ignored flip_z or the 'auto' limits would still pass a test
    that only compared each case with itself."""
    g = golden('renderer_eg3d.npz')
    names = list(CASES)
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            assert max_abs(g[f'{a}/rgb'], g[f'{b}/rgb']) > 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize('case', list(CASES))
def test_product_eg3d_renderer_on_device_vs_reference(golden, case):
    """Tolerance 5e-5 on colours / 2e-4 on depth (fp32 sums in another order than the CPU reference's; BASELINE: 1e-3 on RGB)."""
    g = golden('renderer_eg3d.npz')
    rgb, depth, wsum = _run_product(g, case, 'cuda')
    assert rgb.is_cuda and rgb.shape == g[f'{case}/rgb'].shape
    assert max_abs(rgb.cpu(), g[f'{case}/rgb']) <= 5e-5 and max_abs(wsum.cpu(), g[f'{case}/wsum']) <= 5e-5
    assert max_abs(depth.cpu(), g[f'{case}/depth']) <= 2e-4
