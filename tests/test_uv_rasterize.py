"""Driver-side UV rasteriser: the HIP kernel against the CPU restatement of the pytorch3d algorithm the
reference calls (oracle/uv_rasterize.py; parity unpinned by reference tests, see its header) on synthetic meshes."""
import numpy as np
import pytest
import torch

from oracle import uv_rasterize as OU


def synthetic_mesh(n=24, seed=0, second_layer=True):
    """A bumpy square sheet of 2 n^2 triangles over model-space [-0.15, 0.15]^2 (x 5 = +-0.75 NDC), plus a smaller sheet in front of
    part of it: overlapping depths, silhouettes inside the crop window, uv and mask varying per vertex."""
    rs = np.random.RandomState(seed)

    def sheet(half, z0, n, flip):
        lin = np.linspace(-half, half, n + 1)
        xx, yy = np.meshgrid(lin, lin, indexing='xy')
        v = np.stack([xx, yy, z0 + 0.01 * rs.randn(n + 1, n + 1)], -1).reshape(-1, 3)
        v[:, :2] += 0.2 * half / n * rs.randn(*v[:, :2].shape)
        idx = np.arange((n + 1) ** 2).reshape(n + 1, n + 1)
        a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
        t = np.concatenate([np.stack([a, b, c], 1), np.stack([b, d, c], 1)])
        return v, (t[:, ::-1] if flip else t)
    v1, t1 = sheet(0.15, 0.0, n, False)
    if second_layer:
        v2, t2 = sheet(0.06, -0.05, n // 3, True)          # z flips sign in project(): this sheet ends up nearer or farther consistently
        v2[:, 0] += 0.03
        verts, tris = np.concatenate([v1, v2]), np.concatenate([t1, t2 + len(v1)])
    else:
        verts, tris = v1, t1
    uv = rs.rand(len(verts), 2)
    mask = (rs.rand(len(verts)) > 0.25).astype(np.float32)
    return verts.astype(np.float32), tris.astype(np.int32), uv.astype(np.float32), mask


def test_oracle_rasterizer_known_answers():
    """One triangle: coverage, barycentrics and the image orientation (+X left, +Y up => world x grows with the column)."""
    verts = np.array([[-0.5, -0.5, 0.0], [0.5, -0.5, 0.0], [-0.5, 0.5, 0.0]], np.float32)
    face, bary = OU.rasterize(verts, np.array([[0, 1, 2]]), 8)
    # pixel (r, c) centre is at world ((2c+1)/8 - 1, (2r+1)/8 - 1): inside iff x > -0.5, y > -0.5, x + y < 0; centres exactly ON the
    # hypotenuse are kept too (distance 0 < blur_radius: the blur_radius > 0 rule of the pytorch3d rasteriser)
    xs = (2 * np.arange(8) + 1) / 8 - 1
    want = (xs[None, :] > -0.5) & (xs[:, None] > -0.5) & (xs[None, :] + xs[:, None] < 1e-6)
    assert np.array_equal(face >= 0, want)
    r, c = 2, 3
    w = bary[r, c]
    p = w[0] * verts[0, :2] + w[1] * verts[1, :2] + w[2] * verts[2, :2]
    assert np.allclose(p, [xs[c], xs[r]], atol=1e-6) and abs(w.sum() - 1) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('seed,n', [(0, 24), (1, 40)])
def test_uv_rasterizer_kernel_matches_the_restatement(seed, n):
    from invertavatar_amd.data_preprocess.FaceVerse.renderer import UVRasterizer
    verts, tris, uv, mask = synthetic_mesh(n, seed)
    ras = UVRasterizer(tris, uv, mask, 'cuda')
    tv = ras.project(torch.from_numpy(verts).cuda())
    got = ras.rasterize(tv).cpu().numpy()[0]
    want = OU.make_driven_rendering(tv.cpu().numpy()[0], tris, ras.face_uvcoords.cpu().numpy()[0])
    assert got.shape == (256, 256, 3)
    covered = want[..., 2] > 0
    assert covered.mean() > 0.3                                             # the mesh fills a good part of the crop window
    # face selection / coverage can only differ where two faces are closer in depth than fp32 resolves or a pixel centre sits on the
    # blur boundary: none on these meshes
    assert np.array_equal(got[..., 2], want[..., 2])
    assert np.abs(got - want).max() <= 2e-5
    # batch of two (second mesh shifted): each item equals its own single render
    tv2 = torch.cat([tv, tv + torch.tensor([0.05, -0.03, 0.0], device='cuda')])
    both = ras.rasterize(tv2).cpu().numpy()
    assert np.array_equal(both[0], got)
    want2 = OU.make_driven_rendering(tv2[1].cpu().numpy(), tris, ras.face_uvcoords.cpu().numpy()[0])
    assert np.array_equal(both[1][..., 2], want2[..., 2]) and np.abs(both[1] - want2).max() <= 2e-5


@pytest.mark.gpu
def test_rasterized_condition_drives_the_generator():
    """The rasteriser's output has the contract synthesis() expects: [B,256,256,3], u, v in [-1,1] scaled by the mask, mask in {0,1}."""
    from invertavatar_amd.data_preprocess.FaceVerse.renderer import UVRasterizer
    verts, tris, uv, mask = synthetic_mesh(24, 3, second_layer=False)
    ras = UVRasterizer(tris, uv, np.ones_like(mask), 'cuda')
    img = ras.make_driven_rendering_from_vertices(torch.from_numpy(verts).cuda(), res=256)
    assert img.shape == (1, 256, 256, 3) and set(img[..., 2].unique().tolist()) <= {0.0, 1.0}
    assert img[..., :2].abs().max().item() <= 1.0 + 1e-6
    half = ras.make_driven_rendering_from_vertices(torch.from_numpy(verts).cuda(), res=128)
    assert half.shape == (1, 128, 128, 3) and set(half[..., 2].unique().tolist()) <= {0.0, 1.0}


@pytest.mark.gpu
@pytest.mark.parametrize('res', [128, 192])
def test_uv_rasterizer_resized_output_thresholds_after_the_resize(res):
    """with res != 256 the reference interpolates the continuous mask and thresholds once, afterwards (renderer.py:78-82);
    binarising at 256^2 first moves the silhouette by up to a pixel."""
    from invertavatar_amd.data_preprocess.FaceVerse.renderer import UVRasterizer
    verts, tris, uv, mask = synthetic_mesh(40, 5)
    ras = UVRasterizer(tris, uv, mask, 'cuda')
    tv = ras.project(torch.from_numpy(verts).cuda())
    got = ras.rasterize(tv, res=res).cpu().numpy()[0]
    want = OU.make_driven_rendering(tv.cpu().numpy()[0], tris, ras.face_uvcoords.cpu().numpy()[0], res=res)
    assert got.shape == (res, res, 3)
    assert np.array_equal(got[..., 2], want[..., 2]), int((got[..., 2] != want[..., 2]).sum())
    assert np.abs(got - want).max() <= 2e-5
