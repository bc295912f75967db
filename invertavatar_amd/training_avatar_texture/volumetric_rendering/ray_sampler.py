"""Camera -> ray bundles (reference: volumetric_rendering/ray_sampler.py).

``RaySampler_zxc`` is the sampler the v20 generator uses (ray_sampler.py:65-106):
pixel centres at INTEGER coordinates, K scaled to the render resolution, d = normalize(R K^-1 [i,j,1]).
It is written batched (no Python loop over frames, one 3x3 inverse for the whole batch)."""
import torch

from ... import hipops


class RaySampler(torch.nn.Module):
    """EG3D sampler with half-pixel centres (ray_sampler.py:19-63); API surface for the older generators."""

    def __init__(self):
        super().__init__()
        self.ray_origins_h = self.ray_directions = self.depths = self.image_coords = self.rendering_options = None

    def forward(self, cam2world_matrix, intrinsics, resolution):
        n, dev = cam2world_matrix.shape[0], cam2world_matrix.device
        origin = cam2world_matrix[:, :3, 3]
        fx, fy = intrinsics[:, 0, 0, None], intrinsics[:, 1, 1, None]
        cx, cy, sk = intrinsics[:, 0, 2, None], intrinsics[:, 1, 2, None], intrinsics[:, 0, 1, None]
        centres = (torch.arange(resolution, dtype=torch.float32, device=dev) + 0.5) / resolution
        yy, xx = torch.meshgrid(centres, centres, indexing='ij')
        x_cam = xx.reshape(1, -1).expand(n, -1)
        y_cam = yy.reshape(1, -1).expand(n, -1)
        z_cam = torch.ones_like(x_cam)
        x_lift = (x_cam - cx + cy * sk / fy - sk * y_cam / fy) / fx * z_cam
        y_lift = (y_cam - cy) / fy * z_cam
        pts = torch.stack((x_lift, y_lift, z_cam, torch.ones_like(z_cam)), dim=-1)
        world = torch.bmm(cam2world_matrix, pts.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
        dirs = torch.nn.functional.normalize(world - origin[:, None, :], dim=2)
        return origin.unsqueeze(1).repeat(1, dirs.shape[1], 1), dirs


class RaySampler_zxc(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.ray_origins_h = self.ray_directions = self.depths = self.image_coords = self.rendering_options = None

    def forward(self, cam2world_matrix, cam_K, resolution, normalize=True):
        n, dev = cam2world_matrix.shape[0], cam2world_matrix.device
        if cam2world_matrix.is_cuda and not torch.is_grad_enabled():
            cam25 = torch.cat([cam2world_matrix.reshape(n, 16), cam_K.reshape(n, 9)], 1)
            return hipops.ray_sampler(cam25, resolution, normalize)
        k = cam_K.clone()
        k[:, :2] *= resolution
        k_inv = torch.linalg.inv(k)                                           # [N,3,3]
        pix = torch.linspace(0, resolution - 1, resolution, device=dev)
        jj, ii = torch.meshgrid(pix, pix, indexing='ij')                      # jj: row, ii: column
        homog = torch.stack((ii, jj, torch.ones_like(ii)), -1).reshape(1, -1, 3).expand(n, -1, -1)
        dirs = torch.bmm(homog, k_inv.transpose(1, 2))                        # K^-1 [i, j, 1]
        dirs = torch.bmm(dirs, cam2world_matrix[:, :3, :3].transpose(1, 2))   # rotate into the world frame
        if normalize:
            dirs = torch.nn.functional.normalize(dirs, dim=-1)
        origins = cam2world_matrix[:, None, :3, 3].expand(-1, dirs.shape[1], -1)
        return origins.contiguous(), dirs.contiguous()
