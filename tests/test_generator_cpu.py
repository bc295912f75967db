"""The product's generator modules on CPU tensors (the reference's pure-PyTorch fallback route, BASELINE config 1)
against the golden fixtures: checks the Python mirror (module wiring, names, shapes) without a GPU."""
import os

import pytest
import torch

from invertavatar_amd import synthetic
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
from conftest import GOLDEN, max_abs


@pytest.fixture(scope='module')
def small_generator():
    g = TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)
    return synthetic.fill_parameters(g)


def test_state_dict_names_match_reference_checkpoint_layout():
    """Name-for-name, shape-for-shape compatibility with the reference's pickles."""
    for width, fname in (('small', 'generator_state_names_small.txt'),):
        g = TriPlaneGenerator(**synthetic.generator_kwargs(width))
        mine = {n: (tuple(t.shape), str(t.dtype).replace('torch.', '')) for n, t in g.state_dict().items()}
        ref = {}
        for line in open(os.path.join(GOLDEN, fname)):
            n, s, d = line.rstrip('\n').split('\t')
            ref[n] = (eval(s), d)
        assert set(mine) == set(ref), (sorted(set(ref) - set(mine))[:5], sorted(set(mine) - set(ref))[:5])
        assert mine == ref


def test_full_width_names_and_parameter_count():
    ref = [line.split('\t')[0] for line in open(os.path.join(GOLDEN, 'generator_state_names.txt'))]
    assert len(ref) == 444
    g = TriPlaneGenerator(**synthetic.generator_kwargs('full'))
    assert sorted(g.state_dict().keys()) == sorted(ref)
    assert sum(p.numel() for p in g.parameters()) == 88301010


def test_persistence_and_script_attributes(small_generator):
    g = small_generator
    assert g.init_kwargs['rendering_kwargs']['depth_resolution'] == 48 and g.init_args == ()
    assert (g.z_dim, g.c_dim, g.w_dim, g.img_resolution, g.neural_rendering_resolution) == (512, 25, 512, 512, 128)
    assert g.backbone.mapping.w_avg.shape == (512,)
    g2 = TriPlaneGenerator(*g.init_args, **g.init_kwargs).eval().requires_grad_(False)
    from invertavatar_amd.torch_utils import misc
    misc.copy_params_and_buffers(g, g2, require_all=True)
    assert all(torch.equal(a, b) for a, b in zip(g.state_dict().values(), g2.state_dict().values()))


def test_cpu_synthesis_matches_reference(golden, small_generator):
    gld = golden('generator_small.npz')
    frames, nrr = gld['frames'].tolist(), gld['nrr']
    g = small_generator
    with torch.no_grad():
        ws = g.mapping(synthetic.latent(0, 1), synthetic.conditioning_camera(), truncation_psi=0.7, truncation_cutoff=14)
        assert max_abs(ws, gld['ws'][:1]) <= 1e-5
        ws = gld['ws']
        c, uv = synthetic.camera_labels(frames), synthetic.uv_conditions(frames)
        jit = synthetic.jitter(frames, nrr * nrr)
        out = g.synthesis(ws, c, {'uvcoords_image': uv}, neural_rendering_resolution=nrr, noise_mode='const', evaluation=True,
                          return_featmap=True, jitter=jit)
    assert max_abs(out['feature_image'], gld['feature_image']) <= 5e-5
    assert max_abs(out['image_depth'], gld['image_depth']) <= 5e-5
    assert max_abs(out['triplane'][..., ::4, ::4], gld['triplane_sub4']) <= 5e-5
    assert max_abs(out['image'][:1], gld['image']) <= 1e-4
    with torch.no_grad():
        tex = g.texture_backbone.synthesis(ws, cond_list=None, return_list=True, noise_mode='const')
        sta = g.backbone.synthesis(ws, cond_list=None, return_list=True, noise_mode='const')
        out = g.synthesis_withTexture(ws, tex, c, {'uvcoords_image': uv}, static_feats=sta, neural_rendering_resolution=nrr,
                                      noise_mode='const', evaluation=True, jitter=jit)
    assert max_abs(out['image'][..., ::4, ::4], gld['image_withtexture_sub4']) <= 1e-4


def test_generator_copies_and_pickles_after_a_forward_pass(small_generator):
    """copy.deepcopy / pickle / torch.save of a generator that has already rendered (the calls the reference's scripts make,
    reenact_avatar_next3d.py:158): orchestration state lives outside the module, so only parameters and buffers travel."""
    import copy, io, pickle
    g = small_generator
    frames, nrr = [3], 32
    with torch.no_grad():
        ws = g.mapping(synthetic.latent(0, 1), synthetic.conditioning_camera(), truncation_psi=0.7, truncation_cutoff=14)
        args = (ws, synthetic.camera_labels(frames), {'uvcoords_image': synthetic.uv_conditions(frames)})
        kw = dict(neural_rendering_resolution=nrr, noise_mode='const', evaluation=True, jitter=synthetic.jitter(frames, nrr * nrr))
        ref = g.synthesis(*args, **kw)['image']
        g2 = copy.deepcopy(g)
        g3 = pickle.loads(pickle.dumps(g))
        buf = io.BytesIO()
        torch.save(g, buf)
        buf.seek(0)
        g4 = torch.load(buf, weights_only=False)
        for other in (g2, g3, g4):
            assert torch.equal(other.synthesis(*args, **kw)['image'], ref)
    assert not any(k.startswith('_') and 'stream' in k for k in vars(g))


def _soft_mouth_by_loops(alpha, mouth):
    """fill_mouth's blur_mouth_edge branch (renderer.py:732-736) pixel by pixel: cv2.erode 3x3 three times (border = +inf), cv2.blur 5x5
    (double sums, * double(1/25), -> float; BORDER_REFLECT_101), (255 - x) / 255 in float32."""
    import numpy as np
    h, w = alpha.shape
    img = np.where(mouth == 0, np.float32(255), alpha * np.float32(255)).astype(np.float32)
    for _ in range(3):
        nxt = img.copy()
        for y in range(h):
            for x in range(w):
                nxt[y, x] = img[max(0, y - 1):y + 2, max(0, x - 1):x + 2].min()
        img = nxt
    refl = lambda i, n: -i if i < 0 else (2 * (n - 1) - i if i >= n else i)     # noqa: E731
    out = np.empty_like(img)
    for y in range(h):
        for x in range(w):
            s = 0.0
            for dy in range(-2, 3):
                for dx in range(-2, 3):
                    s += float(img[refl(y + dy, h), refl(x + dx, w)])
            out[y, x] = (np.float32(255) - np.float32(s * (1.0 / 25.0))) / np.float32(255)
    return out


def test_fill_mouth_default_blurs_the_mouth_edge():
    """`fill_mouth(images)` with the signature's default blur_mouth_edge=True used to raise."""
    import numpy as np
    from invertavatar_amd.training_avatar_texture.volumetric_rendering.renderer import fill_mouth
    m = torch.ones(3, 1, 24, 28)
    m[0, 0, 10, 12] = 0                                  # a one-pixel hole in the interior: 7x7 after erosion
    m[1, 0, 1:4, 22:27] = 0                              # a hole next to the top-right corner: reflected border samples
    m[2, 0, 8:16, 6:20] = 0.25 * torch.from_numpy(np.random.RandomState(2).rand(8, 14).astype(np.float32))    # fractional alphas
    full, soft = fill_mouth(m.clone())
    full_hard, hard = fill_mouth(m.clone(), blur_mouth_edge=False)
    assert torch.equal(full, full_hard)                  # the composited alpha uses the unblurred mask (:738)
    for b in range(3):
        expect = _soft_mouth_by_loops(m[b, 0].numpy(), hard[b, 0].numpy())
        assert np.array_equal(soft[b, 0].numpy(), expect), b
    assert soft[0, 0, 10, 12] == 1 and soft[0, 0, 10, 15] == np.float32(1 - np.float32(255 * 10 / 25.0) / 255)   # 3 of 5 columns inside
    assert soft[0, 0, 10, 18] == 0 and soft[0, 0, 4, 12] == 0 and soft[0, 0, 5, 12] == np.float32(0.2)
