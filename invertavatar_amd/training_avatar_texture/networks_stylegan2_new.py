"""Conditioned StyleGAN2 backbone of Next3D++ (reference: training_avatar_texture/networks_stylegan2_new.py).

Relative to the stock network the reference adds exactly three things, all in the forward pass
(SURVEY.md 2.1 row 12), and this module adds exactly those on top of
``training.networks_stylegan2``:

  * ``cond_list``  -- rasterised neural-texture images pasted over the skip image at 32^2 and over
    the feature maps at 32/64/128^2 through their alpha channel (:523-540);
  * ``return_list`` -- the multi-resolution feature list [img@32, x@32, x@64, x@128, x@256, img@256];
  * ``feat_conditions`` -- per-resolution CS-SFT (scale, shift) pairs from the inversion encoder;
and ``mapping_ws`` on the Generator so that the three backbones can share one mapping width.
"""
import numpy as np
import torch

from ..torch_utils import misc, persistence
from ..training import networks_stylegan2 as _base
from ..training.networks_stylegan2 import (normalize_2nd_moment, modulated_conv2d, FullyConnectedLayer, Conv2dLayer,  # noqa: F401
                                           MappingNetwork, SynthesisLayer, ToRGBLayer, SynthesisBlock)


def _paste(cond, x, fused, consumer=None, noise_mode='random', half_ops=False):
    """cond[:, :-1] * a + x * (1 - a), a = cond[:, -1:] (reference :537-540).  `consumer`: the one layer that reads the result (the next
    block's conv0); when it takes split input -- under THIS call's noise mode; ia_cond_blend_split writes the two-plane hi / lo form,
    so not for a consumer that runs with fp16 operands (`half_ops`) -- the blend is written in that format only (hipops.SplitAct)."""
    if fused and cond.dtype == torch.float32 and (x.shape[2] * x.shape[3]) % 4 == 0:
        from invertavatar_amd import hipops
        if (consumer is not None and consumer._pre is not None and x.shape[1] % 8 == 0 and consumer.in_channels == x.shape[1]
                and not half_ops and consumer._takes_split_input(x.shape[2], noise_mode, False)):
            return hipops.cond_blend_split(cond.contiguous(), x.contiguous(), consumer._pre[0], consumer)
        return hipops.cond_blend(cond.contiguous(), x.contiguous())
    a = cond[:, -1:]
    return cond[:, :-1] * a + x * (1 - a)


@persistence.persistent_class
class SynthesisNetwork(_base.SynthesisNetwork):
    def forward_head(self, ws, out_res=(32, 256), **block_kwargs):
        """Blocks 4^2 .. out_res[0]^2, which never see `cond_list` (it is blended in AFTER the out_res[0] block): lets a caller
        start them before the conditioning images exist.  Returns the state to pass as `_head=` to forward()."""
        first = int(np.log2(out_res[0])) - 2
        x = img = None
        self._prepare_styles(ws)
        for idx, (res, cur_ws) in enumerate(zip(self.block_resolutions, self._split_ws(ws))):
            if idx > first:
                break
            x, img = getattr(self, f'b{res}')(x, img, cur_ws, None, _next_conv=self._next_conv(res), _next_half=self._next_half(res, ws, block_kwargs),
                                              **block_kwargs)
        return x, img, first

    def forward(self, ws, cond_list, return_list, feat_conditions=None, return_imgs=False, out_res=(32, 256), _head=None, _tap=None,
                **block_kwargs):
        assert not (return_list and return_imgs)
        first = int(np.log2(out_res[0])) - 2                       # index of the first tapped block (32^2)
        last = (self.img_resolution_log2 - 2) if len(out_res) == 1 else (int(np.log2(out_res[1])) - 2)
        x = img = None
        feats, imgs = [], []
        if _head is None:
            self._prepare_styles(ws)
        for idx, (res, cur_ws) in enumerate(zip(self.block_resolutions, self._split_ws(ws))):
            cond = feat_conditions[res] if (feat_conditions is not None and res in feat_conditions.keys()) else None
            if _head is not None and idx <= _head[2]:
                if idx < _head[2]:
                    continue
                x, img = _head[0], _head[1]                         # resume after the pre-computed head
            else:
                x, img = getattr(self, f'b{res}')(x, img, cur_ws, cond, _next_conv=self._next_conv(res),
                                                  _next_half=self._next_half(res, ws, block_kwargs), **block_kwargs)
            if idx < first:
                continue
            # On the device inference path nothing downstream writes x / img in place (every fused layer allocates its
            # output), so the taps are returned without the reference's defensive clones, and the condition paste
            # (four elementwise kernels in the reference, :537-540) is one launch.
            fused = x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()      # (x: this block's fp32 output)
            if return_list:
                if idx == first:
                    feats.append(img if fused else img.clone())
                feats.append(x if fused else x.clone())
                if _tap is not None and len(feats) in _tap[0]:
                    _tap[1](feats)      # len(feats) taps exist: the caller may let their consumer start (triplane_v20)
            if cond_list is not None:
                if idx == first:   # face region copied straight into the skip image
                    img = _paste(cond_list[0], img, fused)
                if idx < last:     # ... and into the features of the next block's input
                    x = _paste(cond_list[1 + idx - first], x, fused, self._next_conv(res), block_kwargs.get('noise_mode', 'random'),
                               bool(self._next_half(res, ws, block_kwargs)))
        if return_list:
            feats.append(img)
            return feats
        if return_imgs:
            return imgs
        return img


@persistence.persistent_class
class Generator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_ws=-1, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim = z_dim
        self.c_dim = c_dim
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        if mapping_ws == -1:
            mapping_ws = self.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=mapping_ws, **mapping_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
