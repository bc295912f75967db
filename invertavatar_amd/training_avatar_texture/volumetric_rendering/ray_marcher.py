"""MipNeRF-style compositing (reference: volumetric_rendering/ray_marcher.py:25-57).

On the device path of the v20 generator this arithmetic runs inside the fused ``ia_render_rays``
kernel; the module form below is the torch definition used for CPU tensors and by callers that
hold explicit per-sample tensors."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class MipRayMarcher2(nn.Module):
    def __init__(self):
        super().__init__()

    def run_forward(self, colors, densities, depths, rendering_options):
        deltas = depths[:, :, 1:] - depths[:, :, :-1]
        colors_mid = (colors[:, :, :-1] + colors[:, :, 1:]) / 2
        densities_mid = (densities[:, :, :-1] + densities[:, :, 1:]) / 2
        depths_mid = (depths[:, :, :-1] + depths[:, :, 1:]) / 2
        assert rendering_options['clamp_mode'] == 'softplus', 'MipRayMarcher only supports `clamp_mode`=`softplus`!'
        densities_mid = F.softplus(densities_mid - 1)
        alpha = 1 - torch.exp(-(densities_mid * deltas))
        transmittance = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :, :1]), 1 - alpha + 1e-10], -2), -2)[:, :, :-1]
        weights = alpha * transmittance
        composite_rgb = torch.sum(weights * colors_mid, -2)
        weight_total = weights.sum(2)
        composite_depth = torch.sum(weights * depths_mid, -2) / weight_total
        composite_depth = torch.nan_to_num(composite_depth, float('inf'))
        composite_depth = torch.clamp(composite_depth, torch.min(depths), torch.max(depths))
        if rendering_options.get('white_back', False):
            composite_rgb = composite_rgb + 1 - weight_total
        return composite_rgb * 2 - 1, composite_depth, weights

    def forward(self, colors, densities, depths, rendering_options):
        return self.run_forward(colors, densities, depths, rendering_options)
