"""Oracle restatement of the volume renderer.

Functional torch-CPU code; library ops the reference itself calls on CPU
(cumsum / cumprod / searchsorted / sort / softplus) are called directly because their
CPU rounding behaviour is part of the contract.  Test
infrastructure only.
"""
import collections
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops

# Inverse of the three plane-axis matrices, reference: volumetric_rendering/renderer.py:36-46.
# Their effect: plane 0 samples (x, y), plane 1 (x, z), plane 2 (z, x).
_PLANE_AXES = ((0, 1), (0, 2), (2, 0))


def linspace_f32(start, end, steps):
    """CPU torch.linspace bit rule for fp32.

    step = fl32((e-s)/(n-1)); entry k < n/2: fma(step, k, s); else fma(-step, n-1-k, e).
    """
    s = np.float32(start)
    e = np.float32(end)
    step = np.float32((e - s) / np.float32(steps - 1))
    out = np.empty(steps, dtype=np.float32)
    half = steps // 2
    for k in range(steps):
        if k < half:
            out[k] = np.float32(np.float64(step) * k + np.float64(s))
        else:
            out[k] = np.float32(np.float64(e) - np.float64(step) * (steps - 1 - k))
    return torch.from_numpy(out)


def ray_sampler_zxc(cam2world, intrinsics, resolution):
    """Reference: volumetric_rendering/ray_sampler.py:70-107."""
    n = cam2world.shape[0]
    k = intrinsics.clone()
    k[:, :2] *= resolution
    pix = torch.arange(resolution, dtype=torch.float32)
    jj, ii = torch.meshgrid(pix, pix, indexing='ij')           # jj = row (y), ii = col (x)
    homog = torch.stack([ii, jj, torch.ones_like(ii)], -1).reshape(-1, 3)  # [R,3], row-major over (y,x)
    origins, dirs = [], []
    for b in range(n):
        kinv = torch.linalg.inv(k[b])
        d_cam = homog @ kinv.t()
        d = d_cam @ cam2world[b, :3, :3].t()
        d = F.normalize(d, dim=-1)
        dirs.append(d)
        origins.append(cam2world[b, :3, 3].expand_as(d))
    return torch.stack(origins, 0), torch.stack(dirs, 0)


def sample_from_planes(planes, coords, box_warp):
    """planes [B,3,C,H,W], coords [B,M,3] -> [B,3,M,C].

    Reference: volumetric_rendering/renderer.py:51-65,85-97.
    """
    b, p, c, h, w = planes.shape
    co = coords * (2.0 / box_warp)
    feats = []
    for pi, (a0, a1) in enumerate(_PLANE_AXES):
        grid = torch.stack([co[..., a0], co[..., a1]], -1).unsqueeze(1)    # [B,1,M,2]
        s = ops.grid_sample_bilinear(planes[:, pi], grid)                   # [B,C,1,M]
        feats.append(s[:, :, 0].permute(0, 2, 1))
    return torch.stack(feats, 1)


def osg_decoder(dec, feats):
    """feats [B,3,M,32] -> rgb [B,M,32], sigma [B,M,1].

    dec: dict with 'net.0.weight/bias' (64x32), 'net.2.weight/bias' (33x64).
    Reference: training_avatar_texture/triplane_v20.py:415-438.
    """
    x = feats.mean(1)
    b, m, c = x.shape
    x = x.reshape(b * m, c)
    hdn = F.softplus(ops.fully_connected(x, dec['net.0.weight'], dec['net.0.bias'], dec.get('lr_mul', 1.0)))
    o = ops.fully_connected(hdn, dec['net.2.weight'], dec['net.2.bias'], dec.get('lr_mul', 1.0))
    o = o.reshape(b, m, -1)
    rgb = torch.sigmoid(o[..., 1:]) * (1 + 2 * 0.001) - 0.001
    return rgb, o[..., 0:1]


def ray_march(colors, densities, depths, white_back=False):
    """MipRayMarcher2.  Reference: volumetric_rendering/ray_marcher.py:25-57."""
    deltas = depths[:, :, 1:] - depths[:, :, :-1]
    c_mid = (colors[:, :, :-1] + colors[:, :, 1:]) / 2
    d_mid = (densities[:, :, :-1] + densities[:, :, 1:]) / 2
    z_mid = (depths[:, :, :-1] + depths[:, :, 1:]) / 2
    d_mid = F.softplus(d_mid - 1)
    alpha = 1 - torch.exp(-(d_mid * deltas))
    shifted = torch.cat([torch.ones_like(alpha[:, :, :1]), 1 - alpha + 1e-10], -2)
    weights = alpha * torch.cumprod(shifted, -2)[:, :, :-1]
    rgb = torch.sum(weights * c_mid, -2)
    wsum = weights.sum(2)
    depth = torch.sum(weights * z_mid, -2) / wsum
    depth = torch.nan_to_num(depth, float('inf'))
    depth = torch.clamp(depth, torch.min(depths), torch.max(depths))
    if white_back:
        rgb = rgb + 1 - wsum
    return rgb * 2 - 1, depth, weights


def sample_pdf_det(bins, weights, n_importance, eps=1e-5, u=None):
    """Inverse-CDF sampling.  Reference: renderer.py:430-469: deterministic grid (det=True, :450), or -- `u`
    [n_rays, n_importance] given -- the injected uniform draws of the det=False branch (:453, torch.rand).

    Returns samples and the integer buffers (inds, below, above).
    """
    n_rays, n_w = weights.shape
    w = weights + eps
    pdf = w / torch.sum(w, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = torch.linspace(0, 1, n_importance).expand(n_rays, n_importance).contiguous() if u is None else u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, n_w)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    samples = bin_b + (u - cdf_b) / denom * (bin_a - bin_b)
    return samples, inds, below, above, cdf


def smooth_weights(weights):
    """max_pool1d(2,1,pad 1) -> avg_pool1d(2,1) -> +0.01.  Reference: renderer.py:420-423."""
    w = F.max_pool1d(weights.unsqueeze(1), 2, 1, padding=1)
    w = F.avg_pool1d(w, 2, 1).squeeze(1)
    return w + 0.01


def sample_importance(z_vals, weights, n_importance, u=None):
    """Reference: renderer.py:410-428.  z_vals [B,R,S,1], weights [B,R,S-1,1]."""
    b, r, s, _ = z_vals.shape
    z = z_vals.reshape(b * r, s)
    w = smooth_weights(weights.reshape(b * r, -1))
    z_mid = 0.5 * (z[:, :-1] + z[:, 1:])
    samples, inds, below, above, cdf = sample_pdf_det(z_mid, w[:, 1:-1], n_importance, u=u)
    return samples.reshape(b, r, n_importance, 1), dict(inds=inds, below=below, above=above, cdf=cdf)


def coarse_depths(rays_o, n_coarse, jitter):
    """Stratified coarse depths with injected jitter.

    Reference: renderer.py:311-314,404-406.  `jitter` [B,R,S,1] replaces torch.rand_like.
    """
    dist = torch.norm(rays_o, dim=-1).mean().item()
    start, end = dist - 0.45, dist + 0.6
    b, r, _ = rays_o.shape
    lin = torch.linspace(start, end, n_coarse).reshape(1, 1, n_coarse, 1).repeat(b, r, 1, 1)
    delta = (end - start) / (n_coarse - 1)
    return lin + jitter * delta, (start, end)


def render(planes, dec, rays_o, rays_d, jitter, n_coarse=48, n_fine=48, box_warp=1.0, white_back=False,
           return_aux=False, u=None):
    """ImportanceRenderer_bsMotion.forward.  Reference: renderer.py:309-351.  evaluation=True by default; `u` [B*R, n_fine] = the
    uniform draws of evaluation=False (in the order torch.rand returns them: unsorted)."""
    b, r, _ = rays_o.shape
    z_c, (start, end) = coarse_depths(rays_o, n_coarse, jitter)
    xyz = (rays_o.unsqueeze(-2) + z_c * rays_d.unsqueeze(-2)).reshape(b, -1, 3)
    col_c, den_c = osg_decoder(dec, sample_from_planes(planes, xyz, box_warp))
    col_c = col_c.reshape(b, r, n_coarse, -1)
    den_c = den_c.reshape(b, r, n_coarse, 1)
    _, _, w_c = ray_march(col_c, den_c, z_c, white_back)
    z_f, ibuf = sample_importance(z_c, w_c, n_fine, u=u)
    xyz = (rays_o.unsqueeze(-2) + z_f * rays_d.unsqueeze(-2)).reshape(b, -1, 3)
    col_f, den_f = osg_decoder(dec, sample_from_planes(planes, xyz, box_warp))
    col_f = col_f.reshape(b, r, n_fine, -1)
    den_f = den_f.reshape(b, r, n_fine, 1)
    # unify: renderer.py:372-382
    z_all = torch.cat([z_c, z_f], -2)
    col_all = torch.cat([col_c, col_f], -2)
    den_all = torch.cat([den_c, den_f], -2)
    _, order = torch.sort(z_all, dim=-2)
    z_all = torch.gather(z_all, -2, order)
    col_all = torch.gather(col_all, -2, order.expand(-1, -1, -1, col_all.shape[-1]))
    den_all = torch.gather(den_all, -2, order)
    rgb, depth, w = ray_march(col_all, den_all, z_all, white_back)
    if return_aux:
        aux = dict(z_coarse=z_c, w_coarse=w_c, z_fine=z_f, order=order, ray_start=start, ray_end=end,
                   den_coarse=den_c, **ibuf)
        return rgb, depth, w.sum(2), aux
    return rgb, depth, w.sum(2)


def ray_limits_box(rays_o, rays_d, box_side_length):
    """Slab test of every ray against the cube [-s/2, s/2]^3; (-1, -2) marks a miss.
    Reference: volumetric_rendering/math_utils.py:46-98 (the sign-indexed bounds are restated per axis:
    near = bound on the side the ray comes from, far = the other one)."""
    shape = rays_o.shape
    o, d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    half = box_side_length / 2
    inv = 1 / d
    neg = inv < 0
    lo = torch.full_like(o, -half)
    hi = torch.full_like(o, half)
    near = (torch.where(neg, hi, lo) - o) * inv
    far = (torch.where(neg, lo, hi) - o) * inv
    tmin, tmax = near[:, 0], far[:, 0]
    valid = torch.ones_like(tmin, dtype=torch.bool)
    for ax in (1, 2):
        valid &= ~((tmin > far[:, ax]) | (near[:, ax] > tmax))
        tmin = torch.max(tmin, near[:, ax])
        tmax = torch.min(tmax, far[:, ax])
    tmin = torch.where(valid, tmin, torch.full_like(tmin, -1))
    tmax = torch.where(valid, tmax, torch.full_like(tmax, -2))
    return tmin.reshape(*shape[:-1], 1), tmax.reshape(*shape[:-1], 1)


def coarse_depths_eg3d(rays_o, rays_d, ray_start, ray_end, n_coarse, jitter, box_warp):
    """Coarse depths of ImportanceRenderer.  Reference: renderer.py:131-140,220-240.  'auto' limits: per-ray box entry / exit, rays
    that miss get (min, max) of the hitting rays' ENTRY depths (:135-136 -- both from ray_start, as written there); tensor limits
    use math_utils.linspace (:101-118: start + k/(n-1) * (stop - start), NOT torch.linspace's bit rule)."""
    b, r, _ = rays_o.shape
    if ray_start == ray_end == 'auto':
        t0, t1 = ray_limits_box(rays_o, rays_d, box_warp)
        ok = t1 > t0
        if ok.any():
            lo, hi = t0[ok].min(), t0[ok].max()
            t0 = torch.where(ok, t0, lo)
            t1 = torch.where(ok, t1, hi)
        steps = (torch.arange(n_coarse, dtype=torch.float32) / (n_coarse - 1)).reshape(1, 1, n_coarse, 1)
        lin = t0.unsqueeze(-2) + steps * (t1 - t0).unsqueeze(-2)
        return lin + jitter * ((t1 - t0) / (n_coarse - 1)).unsqueeze(-1)
    lin = torch.linspace(ray_start, ray_end, n_coarse).reshape(1, 1, n_coarse, 1).repeat(b, r, 1, 1)
    return lin + jitter * ((ray_end - ray_start) / (n_coarse - 1))


def render_eg3d(planes, dec, rays_o, rays_d, jitter, u, ray_start='auto', ray_end='auto', flip_z=False, n_coarse=48, n_fine=48,
                box_warp=1.0, return_aux=False):
    """ImportanceRenderer.forward.  Reference: renderer.py:122-293.  `jitter` [B,R,S,1] and `u` [B*R, n_fine] are
    the two random draws (:234/:238 torch.rand_like, :453 torch.rand); flip_z negates the z coordinate of every query (:196-197)."""
    b, r, _ = rays_o.shape
    sign = torch.tensor([1.0, 1.0, -1.0 if flip_z else 1.0])

    def decode(z):
        xyz = (rays_o.unsqueeze(-2) + z * rays_d.unsqueeze(-2)).reshape(b, -1, 3) * sign
        col, den = osg_decoder(dec, sample_from_planes(planes, xyz, box_warp))
        return col.reshape(b, r, z.shape[2], -1), den.reshape(b, r, z.shape[2], 1)
    z_c = coarse_depths_eg3d(rays_o, rays_d, ray_start, ray_end, n_coarse, jitter, box_warp)
    col_c, den_c = decode(z_c)
    _, _, w_c = ray_march(col_c, den_c, z_c)
    z_f, ibuf = sample_importance(z_c, w_c, n_fine, u=u)
    col_f, den_f = decode(z_f)
    z_all = torch.cat([z_c, z_f], -2)
    _, order = torch.sort(z_all, dim=-2)
    z_all = torch.gather(z_all, -2, order)
    col_all = torch.gather(torch.cat([col_c, col_f], -2), -2, order.expand(-1, -1, -1, col_c.shape[-1]))
    den_all = torch.gather(torch.cat([den_c, den_f], -2), -2, order)
    rgb, depth, w = ray_march(col_all, den_all, z_all)
    if return_aux:
        return rgb, depth, w.sum(2), dict(z_coarse=z_c, w_coarse=w_c, order=order, **ibuf)
    return rgb, depth, w.sum(2)


def flood_fill_outside(alpha255):
    """4-connected fixed-range flood fill from pixel (0,0): lo 0, up 254, new value 255.

    Restates the cv2.floodFill call at renderer.py:727 (OpenCV 4.6 FLOODFILL_FIXED_RANGE:
    a pixel joins when seed - lo <= value <= seed + up).  Explicit BFS on a numpy array.
    """
    img = np.array(alpha255, dtype=np.float32, copy=True)
    h, w = img.shape
    seed = img[0, 0]
    lo, hi = seed - 0.0, seed + 254.0
    inside = (img >= lo) & (img <= hi)
    seen = np.zeros_like(inside)
    q = collections.deque()
    if inside[0, 0]:
        q.append((0, 0))
        seen[0, 0] = True
    while q:
        y, x = q.popleft()
        img[y, x] = 255.0
        for yy, xx in ((y - 1, x), (y + 1, x), (y, x - 1), (y, x + 1)):
            if 0 <= yy < h and 0 <= xx < w and inside[yy, xx] and not seen[yy, xx]:
                seen[yy, xx] = True
                q.append((yy, xx))
    return img


def fill_mouth(alpha):
    """alpha [B,1,H,W] -> (full_alpha, mouth_mask).  Reference: renderer.py:716-741 with
    blur_mouth_edge=False."""
    masks = []
    for a in alpha:
        filled = flood_fill_outside(a[0].numpy() * 255.0)
        masks.append(torch.from_numpy((255.0 - filled).astype(np.float32))[None] / 255.0)
    mouth = torch.stack(masks, 0)
    return (alpha + mouth).clip(0, 1), mouth
