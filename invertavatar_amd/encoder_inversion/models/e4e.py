"""e4e ("encoder for editing") W+ encoder (reference: encoder_inversion/models/e4e.py:22-134).

IR-SE50 trunk with a 3-level feature pyramid; style 0 is predicted from the coarsest map and every other style is a
delta on top of it (coarse styles from 16^2, middle from 32^2, fine from 64^2)."""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import Module

from ...training.networks_stylegan2 import FullyConnectedLayer
from .helpers import face_pool_to, irse50_trunk, run_trunk
from .layers import Conv2d


class GradualStyleBlock(Module):
    """Stride-2 convs down to 1x1, then an equalised-lr linear layer."""

    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c = out_c
        self.spatial = spatial
        layers = []
        for k in range(int(np.log2(spatial))):
            layers += [Conv2d(in_c if k == 0 else out_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        self.convs = nn.Sequential(*layers)
        self.linear = FullyConnectedLayer(in_features=out_c, out_features=out_c, bias=True, activation='linear', lr_multiplier=1)

    def forward(self, x, xs=None):
        """`xs`: the split-format copy of `x` (hipops.SplitAct) when the caller made one for all heads that read this feature map."""
        if FUSED_HEADS and x.is_cuda:
            from . import trunk_hip
            if trunk_hip.style_head_supported(self, x):
                return self.linear(trunk_hip.style_head_forward(self, x, xs).view(-1, self.out_c))
        return self.linear(self.convs(x).view(-1, self.out_c))


def _upsample_add(x, y):
    """Bilinear (align_corners=True) resize of x to y's size, plus y (e4e.py:48-65); one launch on the device inference path."""
    if x.is_cuda and x.dtype == y.dtype == torch.float32 and not torch.is_grad_enabled():
        from ... import hipops
        return hipops.upsample_bilinear_add(x.contiguous(), y.contiguous())
    return F.interpolate(x, size=y.shape[-2:], mode='bilinear', align_corners=True) + y


class Encoder4Editing(Module):
    def __init__(self, n_styles=18, inp_ch=3):
        super().__init__()
        self.input_layer, self.body = irse50_trunk(inp_ch)
        self.style_count = n_styles
        self.coarse_ind = 3
        self.middle_ind = 7
        self.styles = nn.ModuleList([GradualStyleBlock(512, 512, 16 if i < self.coarse_ind else 32 if i < self.middle_ind else 64)
                                     for i in range(n_styles)])
        self.latlayer1 = Conv2d(256, 512, kernel_size=1, stride=1, padding=0)
        self.latlayer2 = Conv2d(128, 512, kernel_size=1, stride=1, padding=0)

    def _shared_split(self, feats):
        """One split-format copy of a pyramid level for all style heads that read it (device inference path), else None."""
        if not (FUSED_HEADS and feats.is_cuda and feats.dtype == torch.float32 and not torch.is_grad_enabled()):
            return None
        if feats.shape[1] % 8:
            return None
        from ... import hipops
        return hipops.act_split(feats.contiguous())

    def forward(self, x):
        _, (c1, c2, c3) = run_trunk(self.body, self.input_layer(x), (6, 20, 23))
        shared = self._shared_split(c3)
        w = self.styles[0](c3, shared).repeat(self.style_count, 1, 1).permute(1, 0, 2)
        feats = c3
        # The style heads (2 .. 6 stride-2 convolutions on 16^2 .. 1^2 images each, batch 1: ~70 latency-bound launches, 3.8 of the
        # 8.4 ms of an encode) read `feats` and nothing of each other: on the device they go round-robin to a few side streams and
        # their deltas are added afterwards (distinct rows of w: the same sums).
        side = _style_streams(self, x.device) if (STYLE_STREAMS > 1 and x.is_cuda and not torch.is_grad_enabled()) else None
        main = torch.cuda.current_stream(x.device) if side else None
        deltas = {}
        for i in range(1, self.style_count):
            if i == self.coarse_ind:
                feats = p2 = _upsample_add(c3, self.latlayer1(c2))
                shared = self._shared_split(feats)
            elif i == self.middle_ind:
                feats = _upsample_add(p2, self.latlayer2(c1))
                shared = self._shared_split(feats)
            if side is None:
                w[:, i] += self.styles[i](feats, shared)
            else:
                st = side[i % len(side)]
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    deltas[i] = self.styles[i](feats, shared)
                deltas[i].record_stream(main)
                if shared is not None:
                    shared.data.record_stream(st)
        if side is not None:
            for st in side:
                main.wait_stream(st)
            for i, d in deltas.items():
                w[:, i] += d
        return w


FUSED_HEADS = True    # device inference path of GradualStyleBlock: LeakyReLU in the convolutions' epilogues, split format from layer to layer
STYLE_STREAMS = 4     # side streams of the style heads on the device path (1: program order on the caller's stream)


def _style_streams(module, device):
    from ... import _runtime
    st = _runtime.state(module)
    if getattr(st, 'style_streams', None) is None or st.style_streams[0].device != device or len(st.style_streams) != STYLE_STREAMS:
        st.style_streams = [torch.cuda.Stream(device=device) for _ in range(STYLE_STREAMS)]
    return st.style_streams


class e4e(nn.Module):
    """Stand-alone wrapper: encoder + face pooling + latent average of a given generator (e4e.py:136-165)."""

    def __init__(self, n_styles=14, if_load_weights=True, generator=None, set_restyle_encoder=False, **unused):
        super().__init__()
        self.n_styles = n_styles
        self.encoder = self.set_encoder(n_styles, inp_ch=3)
        self.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
        self.generator = generator.train().requires_grad_(False) if generator is not None else None
        self.register_buffer('latent_avg', self.generator.backbone.mapping.w_avg.reshape(1, 512))

    def set_encoder(self, n_styles, inp_ch):
        return Encoder4Editing(n_styles, inp_ch)

    def switch_grad(self, nerf_requires_grad=False):
        for i in range(self.encoder.middle_ind):
            for p in self.encoder.styles[i].parameters():
                p.requires_grad = nerf_requires_grad

    def encode(self, x):
        if x.shape[-1] != 256:
            x = face_pool_to(self.face_pool, x)
        codes = self.encoder(x)
        return codes + self.latent_avg.repeat(codes.shape[0], 1, 1)
