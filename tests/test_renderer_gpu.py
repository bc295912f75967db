"""Parity of the fused ray kernel (ia_render_rays) against the oracle renderer and the reference fixture."""
import numpy as np
import pytest
import torch

from oracle import renderer as OR
from invertavatar_amd import hipops, synthetic
from conftest import rnd, max_abs

pytestmark = pytest.mark.gpu


def _decoder():
    sd = {k: torch.empty(s) for k, s in (('net.0.weight', (64, 32)), ('net.0.bias', (64,)),
                                         ('net.2.weight', (33, 64)), ('net.2.bias', (33,)))}
    return synthetic.fill_parameters(sd, salt=5)


def _run_hip(planes, dec, ro, rd, jit, debug=True):
    dev = 'cuda'
    dist = torch.norm(ro, dim=-1).mean().reshape(1).to(dev)
    return hipops.render_rays(hipops.planes_channels_last(planes.to(dev)), ro.to(dev).contiguous(), rd.to(dev).contiguous(),
                              jit.to(dev).reshape(jit.shape[0], jit.shape[1], 48).contiguous(), dist,
                              dec['net.0.weight'].to(dev), dec['net.0.bias'].to(dev), dec['net.2.weight'].to(dev),
                              dec['net.2.bias'].to(dev), debug=debug)


def test_importance_stage_index_buffers(golden):
    """Given the reference's coarse depths and weights, searchsorted indices and the merge order are identical."""
    g = golden('renderer.npz')
    zc = g['z_coarse'].reshape(-1, 48).cuda().contiguous()
    wc = g['w_coarse'].reshape(-1, 47).cuda().contiguous()
    z_fine, inds, order = hipops.importance_stage(zc, wc)
    ref_inds = g['inds'].reshape(-1, 48)
    # north_star: index buffers are bit-exact.  The kernel reproduces ATen's summation order for the pdf normaliser and the
    # fp64-accumulate / fp32-round cumsum, so there is no tolerance for "near ties".
    mism = int((inds.cpu().long() != ref_inds).sum())
    assert mism == 0, f'{mism} of {ref_inds.numel()} searchsorted indices differ from the reference'
    ref_order = g['order'].reshape(-1, 96)
    assert torch.equal(order.cpu().long(), ref_order), 'merge order differs from the reference sort indices'
    assert max_abs(z_fine.cpu(), g['z_fine'].reshape(-1, 48)) <= 2e-6


def test_fused_renderer_vs_reference_fixture(golden):
    g = golden('renderer.npz')
    frames, nrr = g['frames'].tolist(), g['nrr']
    planes = rnd(20, 2, 3, 32, 64, 64)
    jit = synthetic.jitter(frames, nrr * nrr)
    rgb, depth, wsum, aux = _run_hip(planes, _decoder(), g['rays_o'], g['rays_d'], jit)
    assert max_abs(aux['sigma_coarse'].cpu(), g['den_coarse'].reshape(2, -1, 48)) <= 5e-5
    assert max_abs(aux['w_coarse'].cpu(), g['w_coarse'].reshape(2, -1, 47)) <= 2e-5
    assert max_abs(aux['z_fine'].cpu(), g['z_fine'].reshape(2, -1, 48)) <= 5e-5
    assert max_abs(rgb.cpu(), g['rgb']) <= 1e-4
    assert max_abs(depth.cpu(), g['depth']) <= 1e-4
    assert max_abs(wsum.cpu(), g['wsum']) <= 1e-4


@pytest.mark.parametrize('res,nrr,batch', [(256, 32, 1), (64, 16, 3), (8, 8, 1), ((48, 80), 16, 2)])
def test_fused_renderer_vs_oracle(res, nrr, batch):
    """Square planes take the kernel's shared-axis instantiation, the (48, 80) case the general one."""
    frames = list(range(7, 7 + batch))
    ph, pw = res if isinstance(res, tuple) else (res, res)
    planes = rnd(30 + ph, batch, 3, 32, ph, pw)
    cams = synthetic.camera_labels(frames)
    ro, rd = OR.ray_sampler_zxc(cams[:, :16].view(-1, 4, 4), cams[:, 16:25].view(-1, 3, 3), nrr)
    jit = synthetic.jitter(frames, nrr * nrr)
    dec = _decoder()
    ref_rgb, ref_depth, ref_w = OR.render(planes, dec, ro, rd, jit)
    rgb, depth, wsum = _run_hip(planes, dec, ro, rd, jit, debug=False)
    assert max_abs(rgb.cpu(), ref_rgb) <= 1e-4
    assert max_abs(depth.cpu(), ref_depth) <= 1e-4
    assert max_abs(wsum.cpu(), ref_w) <= 1e-4


def test_channel_major_output_is_the_same_image():
    """ia_render_rays with IA_RENDER_RGB_CHANNEL_MAJOR: the [B,R,32] view equals the default output bit for bit, and its
    permutation to [B,32,nrr,nrr] (what TriPlaneGenerator._render builds) is contiguous without a copy."""
    nrr, frames = 16, [3, 4]
    planes = hipops.planes_channels_last(rnd(77, 2, 3, 32, 64, 64).cuda())
    cams = synthetic.camera_labels(frames)
    ro, rd = OR.ray_sampler_zxc(cams[:, :16].view(-1, 4, 4), cams[:, 16:25].view(-1, 3, 3), nrr)
    ro, rd = ro.cuda().contiguous(), rd.cuda().contiguous()
    jit = synthetic.jitter(frames, nrr * nrr).squeeze(-1).cuda().contiguous()
    dist = torch.norm(ro, dim=-1).mean().reshape(1)
    dec = {k: v.cuda() for k, v in _decoder().items()}
    args = (planes, ro, rd, jit, dist, dec['net.0.weight'], dec['net.0.bias'], dec['net.2.weight'], dec['net.2.bias'])
    a = hipops.render_rays(*args)
    b = hipops.render_rays(*args, channel_major=True)
    assert b[0].shape == a[0].shape and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert b[0].permute(0, 2, 1).reshape(2, 32, nrr, nrr).is_contiguous() and not a[0].permute(0, 2, 1).is_contiguous()


def test_render_rays_is_run_to_run_deterministic():
    """Three launches on the same inputs give the same bits in every output and stage buffer (an experimental decoder variant of
    r02 that mixed fp16 and fp32 MFMAs did not: DESIGN 4.2)."""
    nrr, frames = 32, [5]
    planes = hipops.planes_channels_last(rnd(91, 1, 3, 32, 128, 128).cuda())
    cams = synthetic.camera_labels(frames)
    ro, rd = OR.ray_sampler_zxc(cams[:, :16].view(-1, 4, 4), cams[:, 16:25].view(-1, 3, 3), nrr)
    ro, rd = ro.cuda().contiguous(), rd.cuda().contiguous()
    jit = synthetic.jitter(frames, nrr * nrr).squeeze(-1).cuda().contiguous()
    dist = torch.norm(ro, dim=-1).mean().reshape(1)
    dec = {k: v.cuda() for k, v in _decoder().items()}
    args = (planes, ro, rd, jit, dist, dec['net.0.weight'], dec['net.0.bias'], dec['net.2.weight'], dec['net.2.bias'])
    runs = [hipops.render_rays(*args, debug=True) for _ in range(3)]
    for other in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(runs[0][:3], other[:3]))
        assert all(torch.equal(runs[0][3][k], other[3][k]) for k in runs[0][3])


def test_empty_space_depth_clamp_and_properties():
    """Strongly negative densities: weights vanish, depth is NaN -> +inf -> clamped to the global max sample depth."""
    nrr = 16
    planes = torch.zeros(1, 3, 32, 32, 32)
    cams = synthetic.camera_labels([0])
    ro, rd = OR.ray_sampler_zxc(cams[:, :16].view(-1, 4, 4), cams[:, 16:25].view(-1, 3, 3), nrr)
    jit = synthetic.jitter([0], nrr * nrr)
    dec = _decoder()
    dec['net.2.bias'][0] = -200.0
    ref_rgb, ref_depth, ref_w = OR.render(planes, dec, ro, rd, jit)
    rgb, depth, wsum = _run_hip(planes, dec, ro, rd, jit, debug=False)
    assert ref_w.abs().max() == 0 and wsum.abs().max().item() == 0
    assert max_abs(depth.cpu(), ref_depth) <= 1e-5 and torch.isfinite(depth).all()
    assert max_abs(rgb.cpu(), ref_rgb) <= 1e-6      # all -1


def test_ray_sampler_kernel():
    frames = [0, 33, 111]
    cams = synthetic.camera_labels(frames)
    ro, rd = OR.ray_sampler_zxc(cams[:, :16].view(-1, 4, 4), cams[:, 16:25].view(-1, 3, 3), 64)
    o, d = hipops.ray_sampler(cams.cuda(), 64)
    assert max_abs(o.cpu(), ro) == 0 and max_abs(d.cpu(), rd) <= 2e-6


def test_fused_renderer_with_random_importance_draws_vs_oracle():
    """evaluation=False (the inversion's renders, uvnet.py:180): the importance pass inverts the CDF at uniform draws (renderer.py:453).
    The kernel takes them SORTED; the oracle, like the reference, takes them as drawn and sorts all 96 depths afterwards -- same
    samples, same composite."""
    b, nrr = 2, 16
    frames = [5, 130]
    planes = rnd(21, b, 3, 32, 64, 64)
    c = synthetic.camera_labels(frames)
    ro, rd = OR.ray_sampler_zxc(c[:, :16].reshape(-1, 4, 4), c[:, 16:25].reshape(-1, 3, 3), nrr)
    jit = synthetic.jitter(frames, nrr * nrr)
    u = torch.from_numpy(np.random.RandomState(7).rand(b * nrr * nrr, 48).astype(np.float32))
    dec = _decoder()
    ref_rgb, ref_depth, ref_w = OR.render(planes, dec, ro, rd, jit, u=u)
    dev = 'cuda'
    dist = torch.norm(ro, dim=-1).mean().reshape(1).to(dev)
    rgb, depth, wsum = hipops.render_rays(hipops.planes_channels_last(planes.to(dev)), ro.to(dev).contiguous(), rd.to(dev).contiguous(),
                                          jit.to(dev).reshape(b, nrr * nrr, 48).contiguous(), dist, dec['net.0.weight'].to(dev),
                                          dec['net.0.bias'].to(dev), dec['net.2.weight'].to(dev), dec['net.2.bias'].to(dev),
                                          u_importance=u.sort(dim=-1).values.contiguous().to(dev))
    assert max_abs(rgb.cpu(), ref_rgb) <= 1e-4 and max_abs(wsum.cpu(), ref_w) <= 1e-4
    finite = torch.isfinite(ref_depth)
    assert max_abs(depth.cpu()[finite], ref_depth[finite]) <= 2e-4
    det_rgb, _, _ = hipops.render_rays(hipops.planes_channels_last(planes.to(dev)), ro.to(dev).contiguous(), rd.to(dev).contiguous(),
                                       jit.to(dev).reshape(b, nrr * nrr, 48).contiguous(), dist, dec['net.0.weight'].to(dev),
                                       dec['net.0.bias'].to(dev), dec['net.2.weight'].to(dev), dec['net.2.bias'].to(dev))
    assert max_abs(det_rgb.cpu(), rgb.cpu()) > 1e-4            # the draws are really used


def test_per_frame_dist_batch_equals_one_call_per_frame():
    """IA_RENDER_DIST_PER_FRAME: a batch of frames with every frame's own |ray origin| returns what the script's one call per frame
    returns -- colours, weights AND the depth image, whose clamp range (ray_marcher.py:50) is then each frame's own sample range."""
    frames, nrr, res = [0, 3, 6], 16, 64
    planes = rnd(31, len(frames), 3, 32, res, res)
    planes[1] *= 0.02            # a nearly empty frame: many of its rays clamp to the range limits
    cams = synthetic.camera_labels(frames)
    cams[:, 3] *= torch.tensor([1.0, 1.03, 0.97])          # different distances, so the frames' sample ranges differ
    cams[:, 7] *= torch.tensor([1.0, 1.03, 0.97])
    cams[:, 11] *= torch.tensor([1.0, 1.03, 0.97])
    ro, rd = OR.ray_sampler_zxc(cams[:, :16].view(-1, 4, 4), cams[:, 16:25].view(-1, 3, 3), nrr)
    jit = synthetic.jitter(frames, nrr * nrr).reshape(len(frames), nrr * nrr, 48)
    dec = {k: v.cuda() for k, v in _decoder().items()}
    pl = hipops.planes_channels_last(planes.cuda())
    run = lambda sl, dist: hipops.render_rays(pl[sl].contiguous(), ro[sl].cuda().contiguous(), rd[sl].cuda().contiguous(), jit[sl].cuda().contiguous(),
                                              dist, dec['net.0.weight'], dec['net.0.bias'], dec['net.2.weight'], dec['net.2.bias'])
    dists = torch.norm(ro, dim=-1).mean(dim=1).cuda()
    assert float(dists.max() - dists.min()) > 0.05
    rgb, depth, wsum = run(slice(0, 3), dists.contiguous())
    for k in range(3):
        rgb1, depth1, wsum1 = run(slice(k, k + 1), dists[k:k + 1].contiguous())
        assert torch.equal(rgb[k:k + 1], rgb1) and torch.equal(wsum[k:k + 1], wsum1)
        assert torch.equal(depth[k:k + 1], depth1), f'frame {k}: depth clamp range differs from the one-frame call'
    assert not torch.equal(depth[1].clamp(depth[0].min(), depth[0].max()), depth[1])       # (the frames' ranges really differ)


@pytest.mark.parametrize('planes_out,styled,channel_major', [(2, True, True), (1, True, True), (2, False, False)])
def test_renderer_writes_the_consumer_split_format(golden, planes_out, styled, channel_major):
    """r06: ia_render_rays' second copy of the composited features -- fp16 hi / lo (or one rounded) planes multiplied by the styles of the
    convolution that reads them -- is bit for bit what ia_act_split makes of the fp32 image the same launch wrote."""
    g = golden('renderer.npz')
    frames, nrr = g['frames'].tolist(), g['nrr']
    planes = hipops.planes_channels_last(rnd(20, 2, 3, 32, 64, 64).cuda())
    jit = synthetic.jitter(frames, nrr * nrr).cuda().reshape(2, -1, 48).contiguous()
    ro, rd = g['rays_o'].cuda().contiguous(), g['rays_d'].cuda().contiguous()
    dec = {k: v.cuda() for k, v in _decoder().items()}
    dist = torch.norm(ro, dim=-1).mean().reshape(1)
    styles = (torch.rand(2, 32, device='cuda', generator=torch.Generator(device='cuda').manual_seed(4)) + 0.5) if styled else None
    rgb, _, _ = hipops.render_rays(planes, ro, rd, jit, dist, dec['net.0.weight'], dec['net.0.bias'], dec['net.2.weight'], dec['net.2.bias'],
                                   channel_major=channel_major, split_styles=styles, split_planes=planes_out)
    plain, _, _ = hipops.render_rays(planes, ro, rd, jit, dist, dec['net.0.weight'], dec['net.0.bias'], dec['net.2.weight'], dec['net.2.bias'],
                                     channel_major=channel_major)
    assert torch.equal(rgb, plain) and not hasattr(plain, 'split_data')
    image = rgb.permute(0, 2, 1).reshape(2, 32, nrr, nrr).contiguous()
    want = hipops.act_split(image, styles, planes=planes_out)
    got = rgb.split_data.reshape(2, planes_out, 4, nrr, nrr, 8)
    assert got.shape == want.data.shape and torch.equal(got, want.data)
