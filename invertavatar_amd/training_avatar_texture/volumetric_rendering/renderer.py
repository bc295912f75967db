"""Importance renderer of the tri-plane avatar (reference: volumetric_rendering/renderer.py).

``ImportanceRenderer_bsMotion`` (renderer.py:295-469) is what the v20 generator calls.  For device
tensors with the standard options (48 + 48 samples, softplus clamp, OSG decoder) the whole forward
is ONE launch of ``ia_render_rays``; otherwise the torch definition below runs (that is the
reference's own arithmetic, used for CPU tensors).  ``ImportanceRenderer`` (renderer.py:122-293, the
EG3D class: per-ray box limits or a fixed range, ``flip_z``, drawn importance samples) takes the same
kernel through ``ia_ray_limits_box`` + ``ia_render_rays_box`` under the same rule.  The stratified-sampling noise that the
reference draws with ``torch.rand_like`` even in evaluation mode (renderer.py:406) can be injected
through ``forward(..., jitter=...)`` or ``set_jitter`` so that runs are reproducible.

``fill_mouth`` (renderer.py:716-741) runs on the GPU through ``ia_fill_mouth`` instead of the
reference's device -> host -> cv2.floodFill -> device round trip."""
import math

import numpy as np
import torch
import torch.nn as nn

from .ray_marcher import MipRayMarcher2
from . import math_utils
from ... import hipops


def generate_planes(return_inv=True):
    """Axes of the three feature planes (renderer.py:30-48)."""
    planes = torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                           [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                           [[0, 0, 1], [1, 0, 0], [0, 1, 0]]], dtype=torch.float32)
    return torch.linalg.inv(planes) if return_inv else planes


def project_onto_planes(inv_planes, coordinates):
    """[N,M,3] points -> [N*3,M,2] in-plane coordinates (renderer.py:51-65)."""
    n, m, _ = coordinates.shape
    coords = coordinates.unsqueeze(1).expand(-1, 3, -1, -1).reshape(n * 3, m, 3)
    inv = inv_planes.unsqueeze(0).expand(n, -1, -1, -1).reshape(n * 3, 3, 3)
    return torch.bmm(coords, inv)[..., :2]


def sample_from_planes(inv_planes, plane_features, coordinates, mode='bilinear', padding_mode='zeros', box_warp=None, debug=False):
    """Bilinear tri-plane lookup -> [N, 3, M, C] (renderer.py:85-97)."""
    assert padding_mode == 'zeros'
    n, n_planes, c, h, w = plane_features.shape
    _, m, _ = coordinates.shape
    feats = plane_features.view(n * n_planes, c, h, w)
    grid = project_onto_planes(inv_planes, (2 / box_warp) * coordinates).unsqueeze(1)
    out = torch.nn.functional.grid_sample(feats, grid.float(), mode=mode, padding_mode=padding_mode, align_corners=False)
    return out.permute(0, 3, 2, 1).reshape(n, n_planes, m, c)


def _is_osg_decoder(decoder):
    net = getattr(decoder, 'net', None)
    return (net is not None and len(net) == 3 and tuple(net[0].weight.shape) == (64, 32) and tuple(net[2].weight.shape) == (33, 64)
            and isinstance(net[1], torch.nn.Softplus))


class _RendererBase(nn.Module):
    """Shared torch definition of the coarse / importance / composite pipeline."""

    def run_model(self, planes, decoder, sample_coordinates, sample_directions, options):
        if getattr(self, 'flip_z', False):       # ImportanceRenderer only (renderer.py:196-197); in place, as the reference does it
            sample_coordinates[..., -1] *= -1
        feats = sample_from_planes(self.plane_axes.clone(), planes, sample_coordinates, padding_mode='zeros', box_warp=options['box_warp'])
        out = decoder(feats, sample_directions)
        if options.get('density_noise', 0) > 0:
            out['sigma'] += torch.randn_like(out['sigma']) * options['density_noise']
        return out

    def sort_samples(self, all_depths, all_colors, all_densities):
        _, idx = torch.sort(all_depths, dim=-2)
        return (torch.gather(all_depths, -2, idx), torch.gather(all_colors, -2, idx.expand(-1, -1, -1, all_colors.shape[-1])),
                torch.gather(all_densities, -2, idx.expand(-1, -1, -1, 1)))

    def unify_samples(self, depths1, colors1, densities1, depths2, colors2, densities2):
        return self.sort_samples(torch.cat([depths1, depths2], dim=-2), torch.cat([colors1, colors2], dim=-2),
                                 torch.cat([densities1, densities2], dim=-2))

    def sample_stratified(self, ray_origins, ray_start, ray_end, depth_resolution, disparity_space_sampling=False, jitter=None):
        """Evenly spaced depths + one stratum of noise (renderer.py:384-408)."""
        n, m, _ = ray_origins.shape
        dev = ray_origins.device

        def noise(like):
            return torch.rand_like(like) if jitter is None else jitter.to(like.dtype).reshape(like.shape)
        if disparity_space_sampling:
            d = torch.linspace(0, 1, depth_resolution, device=dev).reshape(1, 1, depth_resolution, 1).repeat(n, m, 1, 1)
            d = d + noise(d) * (1 / (depth_resolution - 1))
            return 1. / (1. / ray_start * (1. - d) + 1. / ray_end * d)
        if type(ray_start) == torch.Tensor:
            d = math_utils.linspace(ray_start, ray_end, depth_resolution).permute(1, 2, 0, 3)
            delta = (ray_end - ray_start) / (depth_resolution - 1)
            return d + noise(d) * delta[..., None]
        d = torch.linspace(ray_start, ray_end, depth_resolution, device=dev).reshape(1, 1, depth_resolution, 1).repeat(n, m, 1, 1)
        return d + noise(d) * ((ray_end - ray_start) / (depth_resolution - 1))

    def sample_importance(self, z_vals, weights, N_importance, det=False, u=None):
        """Smoothed inverse-CDF resampling (renderer.py:410-428); note 47 bins but 45 weights are used."""
        with torch.no_grad():
            b, r, s, _ = z_vals.shape
            z = z_vals.reshape(b * r, s)
            w = weights.reshape(b * r, -1)
            w = torch.nn.functional.max_pool1d(w.unsqueeze(1).float(), 2, 1, padding=1)
            w = torch.nn.functional.avg_pool1d(w, 2, 1).squeeze(1) + 0.01
            z_mid = 0.5 * (z[:, :-1] + z[:, 1:])
            return self.sample_pdf(z_mid, w[:, 1:-1], N_importance, det, u=u).detach().reshape(b, r, N_importance, 1)

    def sample_pdf(self, bins, weights, N_importance, det=False, eps=1e-5, u=None):
        """renderer.py:430-469."""
        n_rays, n_w = weights.shape
        weights = weights + eps
        pdf = weights / torch.sum(weights, -1, keepdim=True)
        cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
        if det:
            u = torch.linspace(0, 1, N_importance, device=bins.device).expand(n_rays, N_importance)
        elif u is None:
            u = torch.rand(n_rays, N_importance, device=bins.device)
        else:
            u = u.to(device=bins.device, dtype=bins.dtype).reshape(n_rays, N_importance)      # (the caller's draws in place of :453's)
        u = u.contiguous()
        inds = torch.searchsorted(cdf, u, right=True)
        below, above = torch.clamp_min(inds - 1, 0), torch.clamp_max(inds, n_w)
        cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
        bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
        denom = cdf_a - cdf_b
        denom = torch.where(denom < eps, torch.ones_like(denom), denom)
        return bin_b + (u - cdf_b) / denom * (bin_a - bin_b)

    def _two_pass(self, planes, decoder, ray_origins, ray_directions, depths_coarse, options, det, u_importance=None):
        b, r, s, _ = depths_coarse.shape
        xyz = (ray_origins.unsqueeze(-2) + depths_coarse * ray_directions.unsqueeze(-2)).reshape(b, -1, 3)
        dirs = ray_directions.unsqueeze(-2).expand(-1, -1, s, -1).reshape(b, -1, 3)
        out = self.run_model(planes, decoder, xyz, dirs, options)
        col_c = out['rgb'].reshape(b, r, s, out['rgb'].shape[-1])
        den_c = out['sigma'].reshape(b, r, s, 1)
        n_imp = options['depth_resolution_importance']
        if n_imp <= 0:
            rgb, depth, weights = self.ray_marcher(col_c, den_c, depths_coarse, options)
            return rgb, depth, weights.sum(2)
        _, _, weights = self.ray_marcher(col_c, den_c, depths_coarse, options)
        depths_fine = self.sample_importance(depths_coarse, weights, n_imp, det=det, u=None if det else u_importance)
        xyz = (ray_origins.unsqueeze(-2) + depths_fine * ray_directions.unsqueeze(-2)).reshape(b, -1, 3)
        dirs = ray_directions.unsqueeze(-2).expand(-1, -1, n_imp, -1).reshape(b, -1, 3)
        out = self.run_model(planes, decoder, xyz, dirs, options)
        col_f = out['rgb'].reshape(b, r, n_imp, out['rgb'].shape[-1])
        den_f = out['sigma'].reshape(b, r, n_imp, 1)
        z, c, d = self.unify_samples(depths_coarse, col_c, den_c, depths_fine, col_f, den_f)
        rgb, depth, weights = self.ray_marcher(c, d, z, options)
        return rgb, depth, weights.sum(2)


class ImportanceRenderer(_RendererBase):
    """EG3D renderer with box-limited rays and stochastic importance samples (renderer.py:122-293)."""

    def __init__(self, flip_z=False):
        super().__init__()
        self.ray_marcher = MipRayMarcher2()
        self.plane_axes = generate_planes()
        self.flip_z = flip_z

    def _fused_ok(self, planes, decoder, options):
        """Device tensors with the standard options: the whole forward is ia_ray_limits_box (+ repair) and ONE ia_render_rays_box launch."""
        auto = options['ray_start'] == options['ray_end'] == 'auto'
        return (planes.is_cuda and planes.dtype == torch.float32 and planes.shape[1] == 3 and planes.shape[2] == 32
                and options['depth_resolution'] == 48 and options['depth_resolution_importance'] == 48
                and not options['disparity_space_sampling'] and options.get('clamp_mode') == 'softplus'
                and options.get('density_noise', 0) == 0 and _is_osg_decoder(decoder) and not torch.is_grad_enabled()
                and (auto or not any(isinstance(options[k], (str, torch.Tensor)) for k in ('ray_start', 'ray_end'))))

    def _forward_fused(self, planes, decoder, ray_origins, ray_directions, options):
        b, r, _ = ray_origins.shape
        dev = planes.device
        ro, rd = ray_origins.float().contiguous(), ray_directions.float().contiguous()
        limits, start, end = None, 0.0, 0.0
        if options['ray_start'] == options['ray_end'] == 'auto':
            limits = hipops.ray_limits_box(ro, rd, options['box_warp'], repair_misses=True)
        else:
            start, end = float(options['ray_start']), float(options['ray_end'])
        # the two draws in the reference's order and call shapes: rand_like(depths_coarse) (:238/:242), then rand(n_rays, N_importance)
        # (:280); the importance draws go to the kernel sorted (monotone inverse CDF: same set of fine samples, same merged order)
        jitter = torch.rand_like(torch.empty((b, r, 48, 1), device=dev)).to(torch.float32).reshape(b, r, 48).contiguous()
        u = torch.rand(b * r, 48, device=dev).to(device=dev, dtype=torch.float32).sort(dim=-1).values.contiguous()
        planes_cl = planes.permute(0, 1, 3, 4, 2)
        if not planes_cl.is_contiguous():
            planes_cl = planes_cl.contiguous()
        net = decoder.net
        return hipops.render_rays_box(planes_cl, ro, rd, jitter, u, net[0].weight.detach(), net[0].bias.detach(), net[2].weight.detach(),
                                      net[2].bias.detach(), ray_limits=limits, ray_start=start, ray_end=end, flip_z=bool(self.flip_z),
                                      lr_multiplier=float(net[0].bias_gain), box_warp=options['box_warp'],
                                      white_back=options.get('white_back', False))

    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options):
        if self._fused_ok(planes, decoder, rendering_options):
            return self._forward_fused(planes, decoder, ray_origins, ray_directions, rendering_options)
        self.plane_axes = self.plane_axes.to(ray_origins.device)
        if rendering_options['ray_start'] == rendering_options['ray_end'] == 'auto':
            ray_start, ray_end = math_utils.get_ray_limits_box(ray_origins, ray_directions, box_side_length=rendering_options['box_warp'])
            is_ray_valid = ray_end > ray_start
            if torch.any(is_ray_valid).item():
                ray_start[~is_ray_valid] = ray_start[is_ray_valid].min()
                ray_end[~is_ray_valid] = ray_start[is_ray_valid].max()
        else:
            ray_start, ray_end = rendering_options['ray_start'], rendering_options['ray_end']
        depths = self.sample_stratified(ray_origins, ray_start, ray_end, rendering_options['depth_resolution'],
                                        rendering_options['disparity_space_sampling'])
        return self._two_pass(planes, decoder, ray_origins, ray_directions, depths, rendering_options, det=False)


class ImportanceRenderer_bsMotion(_RendererBase):
    def __init__(self):
        super().__init__()
        self.ray_marcher = MipRayMarcher2()
        self.plane_axes = generate_planes()
        self._jitter = None

    def set_jitter(self, jitter):
        """Use `jitter` ([B,R,48(,1)] in [0,1)) instead of fresh uniform noise for the NEXT forward call."""
        self._jitter = jitter

    def _fused_ok(self, planes, decoder, ray_origins, options):
        return (planes.is_cuda and planes.dtype == torch.float32 and planes.shape[1] == 3 and planes.shape[2] == 32
                and options['depth_resolution'] == 48 and options['depth_resolution_importance'] == 48
                and not options['disparity_space_sampling'] and options.get('clamp_mode') == 'softplus'
                and options.get('density_noise', 0) == 0 and _is_osg_decoder(decoder) and not torch.is_grad_enabled())

    def forward(self, planes, decoder, ray_origins, ray_directions, rendering_options, evaluation=False, jitter=None, dist=None,
                u_importance=None, split_styles=None, split_planes=0):
        """`u_importance` ([B*R, 48] uniform draws, any order) replaces the importance pass's own torch.rand (:453) when the call is
        not `evaluation`: callers that shard a stochastic render over ranks hand every frame the draws it gets in the one-process call.
        `dist` overrides the batch mean of |ray origin| (:311).  One element: a sharded batch passes the value of the whole batch
        so that the depth range does not depend on the sharding.  `split_planes` (1 | 2, fused device route only): the composited features also
        come back in the operand format of the convolution that reads them, multiplied by its `split_styles` [B,32], as the attribute
        `split_data` of the first result (hipops.render_rays).  B elements: frame b uses dist[b] -- a batch of frames that the
        caller's script renders one call each (eval_seq.py:206-212) keeps the per-call results, the depth image included (its clamp
        range, ray_marcher.py:50, is then every frame's own sample range: tests/test_renderer_gpu.py)."""
        if jitter is None:
            jitter, self._jitter = self._jitter, None
        b, r, _ = ray_origins.shape
        per_frame = dist is not None and dist.numel() == b and b > 1
        if dist is not None and not per_frame and dist.numel() != 1:
            raise ValueError(f'dist must have 1 or B = {b} elements, got {dist.numel()}')
        n_coarse = rendering_options['depth_resolution']
        if self._fused_ok(planes, decoder, ray_origins, rendering_options):
            # evaluation: deterministic importance grid.  Otherwise (the inversion's renders, uvnet.py:180) the uniform draws of
            # sample_pdf (:453, same call shape), handed to the kernel sorted: the inverse CDF is monotone, so the set of fine samples
            # and, ties aside, the merged order are those of the reference's unsorted draws + torch.sort.
            if jitter is None:
                # same call shape as the reference's torch.rand_like(depths_coarse) (renderer.py:406)
                jitter = torch.rand_like(torch.empty((b, r, n_coarse, 1), device=planes.device))
            jitter = jitter.to(device=planes.device, dtype=torch.float32).reshape(b, r, n_coarse).contiguous()
            u_imp = None
            if not evaluation:      # (drawn after the jitter, as the reference's generator sees the two calls)
                n_imp = rendering_options['depth_resolution_importance']
                u_imp = torch.rand(b * r, n_imp, device=planes.device) if u_importance is None else u_importance.to(planes.device).reshape(b * r, n_imp)
                u_imp = u_imp.sort(dim=-1).values.to(torch.float32).contiguous()
            if dist is None:
                dist = torch.norm(ray_origins, dim=-1).mean().reshape(1)  # stays on the device: no host sync
            dist = dist.to(device=planes.device, dtype=torch.float32).reshape(-1).contiguous()
            lr_mul = float(decoder.net[0].bias_gain)
            planes_cl = planes.permute(0, 1, 3, 4, 2)          # free when the planes already live channels-last
            if not planes_cl.is_contiguous():
                planes_cl = planes_cl.contiguous()
            return hipops.render_rays(planes_cl, ray_origins.contiguous(), ray_directions.contiguous(),
                                      jitter, dist, decoder.net[0].weight.detach(), decoder.net[0].bias.detach(),
                                      decoder.net[2].weight.detach(), decoder.net[2].bias.detach(), lr_multiplier=lr_mul,
                                      box_warp=rendering_options['box_warp'], white_back=rendering_options.get('white_back', False),
                                      channel_major=True,     # [B,R,32] view of a [B,32,R] image: the caller's permute is free
                                      u_importance=u_imp, split_styles=split_styles, split_planes=split_planes)
        # torch definition (CPU tensors, training, non-standard options)
        if per_frame:      # one reference-shaped call per frame (the depth clamp bounds are then per frame too, as in those calls)
            parts = [self.forward(planes[k:k + 1], decoder, ray_origins[k:k + 1], ray_directions[k:k + 1], rendering_options, evaluation,
                                  None if jitter is None else jitter[k:k + 1], dist.reshape(-1)[k:k + 1],
                                  None if u_importance is None else u_importance.reshape(b, r, -1)[k]) for k in range(b)]
            return tuple(torch.cat(t, 0) for t in zip(*parts))
        self.plane_axes = self.plane_axes.to(ray_origins.device)
        dist = torch.norm(ray_origins, dim=-1).mean().item() if dist is None else float(dist.reshape(-1)[0].item())
        depths = self.sample_stratified(ray_origins, dist - 0.45, dist + 0.6, n_coarse, rendering_options['disparity_space_sampling'],
                                        jitter=jitter)
        return self._two_pass(planes, decoder, ray_origins, ray_directions, depths, rendering_options, det=evaluation, u_importance=u_importance)


def fill_mouth(images, blur_mouth_edge=True):
    """Close the mouth hole of a rasterised face mask: every background region NOT connected to the image
    corner becomes foreground.  images [B,1,H,W] in {0,1}.  Returns (filled alpha, mouth mask) (renderer.py:716-741).
    With ``blur_mouth_edge`` (the signature's default; the v20 generator passes False, triplane_v20.py:74,323) the returned mask is
    eroded three times and box-blurred (:732-736).  Device tensors: ``ia_fill_mouth`` (+ ``ia_mouth_edge_blur``); CPU tensors: breadth-first
    flood fill in NumPy (+ torch pooling)."""
    if images.is_cuda:
        alpha = images.float().contiguous()
        mouth = hipops.fill_mouth(alpha)
        soft = hipops.mouth_edge_blur(alpha, mouth) if blur_mouth_edge else mouth
    else:
        mouth = torch.stack([_flood_fill_cpu(img[0]) for img in images], 0).unsqueeze(1)
        soft = _erode_blur_cpu(images.float(), mouth) if blur_mouth_edge else mouth
    return (images + mouth).clip(0, 1), soft


def _erode_blur_cpu(alpha, mouth):
    """`cv2.blur(cv2.erode(filled, ones(3,3), iterations=3), (5,5))` then (255 - .)/255 (renderer.py:732-736) in torch: three 3x3
    erosions = one 7x7 minimum (border = +inf), normalised 5x5 box filter with BORDER_REFLECT_101 summed in double and scaled by
    the double 1/25 before rounding to float (cv::boxFilter on CV_32F).  OpenCV is not in this image: semantics restated."""
    f = torch.nn.functional
    filled = torch.where(mouth == 0, torch.full_like(alpha, 255.0), alpha * 255.0)
    eroded = -f.max_pool2d(-filled, kernel_size=7, stride=1, padding=3)          # (max_pool2d pads with -inf: the border never wins)
    padded = f.pad(eroded.double(), (2, 2, 2, 2), mode='reflect')                 # 'reflect' = BORDER_REFLECT_101
    box = f.avg_pool2d(padded, kernel_size=5, stride=1, divisor_override=1)
    return (255.0 - (box * (1.0 / 25.0)).float()) / 255.0


def _flood_fill_cpu(alpha):
    """4-connected fixed-range flood from pixel (0,0): lo 0, up 254 on alpha*255; returns (255 - filled)/255."""
    import collections
    img = (alpha.detach().cpu().numpy().astype(np.float32) * 255.0)
    h, w = img.shape
    passable = (img >= img[0, 0]) & (img <= img[0, 0] + 254.0)
    seen = np.zeros((h, w), bool)
    queue = collections.deque([(0, 0)])
    seen[0, 0] = True
    while queue:
        y, x = queue.popleft()
        img[y, x] = 255.0
        for yy, xx in ((y - 1, x), (y + 1, x), (y, x - 1), (y, x + 1)):
            if 0 <= yy < h and 0 <= xx < w and passable[yy, xx] and not seen[yy, xx]:
                seen[yy, xx] = True
                queue.append((yy, xx))
    return torch.from_numpy((255.0 - img) / 255.0).to(alpha.device)
