"""Interval compositing of per-sample colours and densities (API of the reference's
volumetric_rendering/ray_marcher.py:25-57; semantics restated in SURVEY.md Appendix C9).

On the device path of the v20 generator this arithmetic runs inside the fused ``ia_render_rays`` kernel.  The module below
is the torch formulation for CPU tensors and for callers that hold explicit per-sample tensors ([B, R, S, C] colours,
[B, R, S, 1] densities and depths, S samples sorted by depth)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _interval_means(t):
    """Mean of each pair of neighbouring samples along the sample axis: S samples -> S - 1 intervals."""
    return 0.5 * (t[:, :, 1:] + t[:, :, :-1])


def _interval_weights(sigma, dt):
    """Compositing weight of every interval: opacity a_i = 1 - exp(-sigma_i * dt_i) seen through the transparency
    prod_{j<i} (1 - a_j + 1e-10) of the intervals in front of it."""
    opacity = 1 - torch.exp(-(sigma * dt))
    see_through = torch.cumprod(1 - opacity + 1e-10, dim=-2)
    in_front = torch.cat([torch.ones_like(see_through[:, :, :1]), see_through[:, :, :-1]], dim=-2)
    return opacity * in_front


class MipRayMarcher2(nn.Module):
    def __init__(self):
        super().__init__()

    def run_forward(self, colors, densities, depths, rendering_options):
        if rendering_options['clamp_mode'] != 'softplus':
            raise AssertionError('MipRayMarcher only supports `clamp_mode`=`softplus`!')
        sigma = F.softplus(_interval_means(densities) - 1)          # activation bias of -1 as in the reference
        weights = _interval_weights(sigma, depths[:, :, 1:] - depths[:, :, :-1])
        total = weights.sum(2)
        rgb = (weights * _interval_means(colors)).sum(-2)
        # expected depth; rays that hit nothing give 0/0 -> +inf -> the far end of the whole tensor's depth range
        depth = (weights * _interval_means(depths)).sum(-2) / total
        depth = torch.nan_to_num(depth, float('inf')).clamp(depths.min(), depths.max())
        if rendering_options.get('white_back', False):
            rgb = rgb + 1 - total
        return 2 * rgb - 1, depth, weights

    def forward(self, colors, densities, depths, rendering_options):
        return self.run_forward(colors, densities, depths, rendering_options)
