"""H1 harness (invertavatar_amd.reenact_avatar_next3d) against outputs of the reference's own script helpers and its frame loop
(tests/golden/make_golden.py:gen_harness), on CPU tensors: layout_grid modes, parse helpers, seed -> w chain, mosaics."""
import numpy as np
import torch

from invertavatar_amd import reenact_avatar_next3d as R, synthetic
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator


def test_layout_grid_and_parsers_match_the_script(golden):
    g = golden('harness.npz')
    x = g['grid_in']
    assert np.array_equal(R.layout_grid(x, grid_w=3, grid_h=2), g['grid_3x2_hwc'].numpy())
    assert np.array_equal(R.layout_grid(x, grid_w=None, grid_h=1, chw_to_hwc=False), g['grid_6x1_chw'].numpy())
    assert np.array_equal(R.layout_grid(x, grid_w=1, grid_h=6), g['grid_1x6_hwc'].numpy())
    assert R.parse_range('1,2,5-7') == [1, 2, 5, 6, 7] and R.parse_range([4]) == [4]
    assert R.parse_tuple('4x2') == (4, 2) and R.parse_tuple('0,1') == (0, 1) and R.parse_tuple((1, 2)) == (1, 2)


def run_harness(gld, device):
    G = TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)
    synthetic.fill_parameters(G)
    G = G.to(device)
    seeds, frames, nrr = gld['seeds'].tolist(), gld['frames'].tolist(), gld['nrr']
    ws, cond = R.seed_latents(G, seeds, truncation_psi=gld['psi'], truncation_cutoff=gld['cutoff'])

    class Drive(R.SyntheticDrive):     # the fixture's drive frames are orbit frames 5 and 60
        def __getitem__(self, k):
            item = R.SyntheticDrive(0, nrr, True).__getitem__(frames[k])
            return item
    G.neural_rendering_resolution = nrr
    mosaics = R.run_video_animation(G, Drive(len(frames), nrr, True), seeds, grid_dims=(3, 1), truncation_psi=gld['psi'],
                                    truncation_cutoff=gld['cutoff'], neural_rendering_resolution=nrr)
    return torch.cat(ws).cpu(), cond.cpu(), np.stack(mosaics)


def check_mosaics(gld, ws, cond, m, max_off_by_one=2e-3):
    assert (cond - gld['cond']).abs().max().item() <= 1e-6
    assert (ws - gld['ws']).abs().max().item() <= 1e-5
    ref4, refc = gld['mosaic_sub4'].numpy().astype(np.int16), gld['mosaic_crop'].numpy().astype(np.int16)
    d4 = np.abs(m[:, ::4, ::4].astype(np.int16) - ref4)
    dc = np.abs(m[:, 192:320, 512 + 192:512 + 320].astype(np.int16) - refc)
    # uint8 truncation turns a 1e-5 float difference into an off-by-one where x*127.5+128 sits on an integer: at most one
    # level, in a small fraction of the bytes
    assert d4.max() <= 1 and dc.max() <= 1
    assert (d4 > 0).mean() <= max_off_by_one and (dc > 0).mean() <= max_off_by_one
    assert np.abs(m.astype(np.int64).sum(axis=(1, 2)) - gld['mosaic_sum'].numpy()).max() <= max_off_by_one * m[0].size


def test_reenactment_loop_matches_the_script_on_cpu(golden):
    gld = golden('harness.npz')
    check_mosaics(gld, *run_harness(gld, 'cpu'))
