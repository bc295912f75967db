"""UNet encoders with ConvGRU decoders (reference: encoder_inversion/models/unet_encoders.py).

``TriPlanefeat_Encoder`` (:101-246) predicts offsets for the first four neural-texture features;
``TriPlaneSFTfeat_Encoder`` (:249-362) predicts CS-SFT (scale, shift) pairs for the static backbone at 16..256^2.
Both share an IR-SE50 trunk (taps after units 2 / 6 / 20 / 21) and four PixelShuffle up-stages; with ``use_gru`` each
stage ends in a ConvGRU that integrates over the T source frames of a group and carries its state across groups.
Note for callers: the reference runs these decoders with BatchNorm in TRAIN mode at evaluation time
(eval_seq.py:92,96-97) -- batch statistics over the T frames; module modes are the caller's to set, as there."""
import numpy as np
import torch
from torch import nn
from torch.nn import Module
import torch.nn.functional as F

from .helpers import face_pool_to, irse50_trunk, run_trunk
from .layers import Conv2d


HIP_GRU_CONVS = True      # ConvGRU cells, DoubleConv and CS-SFT heads: their convolutions through ia_conv2d_mfma_sx instead of the library


class ConvGRU(torch.nn.Module):
    """r, z = sigmoid(conv([x, h]));  c = tanh(conv([x, r*h]));  h' = (1-z) h + z c   (:8-49)."""

    def __init__(self, channels: int, kernel_size: int = 3, padding: int = 1, out_act_prelu=False):
        super().__init__()
        self.channels = channels
        self.ih = torch.nn.Sequential(Conv2d(channels * 2, channels * 2, kernel_size, padding=padding), torch.nn.Sigmoid())
        self.hh = torch.nn.Sequential(Conv2d(channels * 2, channels, kernel_size, padding=padding),
                                      nn.PReLU(channels) if out_act_prelu else torch.nn.Tanh())

    def _fused(self, x):
        """Device inference path: the convolutions through the library, everything between them in two HIP launches per step."""
        return (x.is_cuda and x.dtype == torch.float32 and (x.shape[-1] * x.shape[-2]) % 4 == 0
                and not (torch.is_grad_enabled() and (x.requires_grad or self.ih[0].weight.requires_grad)))

    def _hip_convs(self, x):
        """The cell's two 3x3 convolutions on ia_conv2d_mfma_sx (fp32 products from fp16 hi / lo pairs) instead of the library:
        the sizes trunk_hip.sx_size_ok names (8^2 up for the cells of the model), channel counts in units of 8.  Packed weights are
        cached per cell."""
        conv_ih, conv_hh = self.ih[0], self.hh[0]
        c2, h, w = conv_ih.in_channels, x.shape[-2], x.shape[-1]
        plain = all(c.kernel_size == (3, 3) and c.padding == (1, 1) and c.stride == (1, 1) and c.dilation == (1, 1) and c.groups == 1
                    and c.padding_mode == 'zeros' and c.bias is not None for c in (conv_ih, conv_hh))
        from . import trunk_hip
        if not (HIP_GRU_CONVS and plain and c2 % 16 == 0 and conv_hh.in_channels == c2 and trunk_hip.sx_size_ok(c2, c2, h, w)
                and trunk_hip.sx_size_ok(c2, self.channels, h, w)):
            return None
        from ... import _runtime, hipops
        st = _runtime.state(self)
        key = tuple((t.data_ptr(), t._version) for t in (conv_ih.weight, conv_hh.weight)) + (conv_ih.weight.device,)
        if getattr(st, 'gru_key', None) != key:
            st.gru_w = (hipops.pack_conv_weight_split(conv_ih.weight.detach().float()), hipops.pack_conv_weight_split(conv_hh.weight.detach().float()))
            st.gru_key = key
        return st.gru_w

    def _step_fused(self, xh, x, h, x_next):
        """xh = cat[x, h] (made by the previous step's update launch: fp32, or a hipops.SplitAct when the cell's convolutions run on
        ia_conv2d_mfma_sx).  Returns (h', cat[x_next, h'] in the same form or None)."""
        from ... import hipops
        conv_ih, conv_hh, act = self.ih[0], self.hh[0], self.hh[1]
        packed = self._hip_convs(x)
        prelu_w = act.weight.detach().float().contiguous() if isinstance(act, nn.PReLU) else None
        if packed is not None and self.channels % 8 == 0 and (x.shape[-2] * x.shape[-1]) % 4 == 0:      # (the split launches walk the pixels in fours)
            # the element-wise launches write the convolutions' operand format themselves (no fp32 cat, no ia_act_split: 4 launches per
            # step between the two convolutions and their fix-ups instead of 6)
            xs = xh if isinstance(xh, hipops.SplitAct) else hipops.act_split(xh)
            gates_pre = hipops.conv2d_mfma_sx(xs, packed[0], bias=conv_ih.bias.detach().float())
            cand_pre = hipops.conv2d_mfma_sx(hipops.convgru_gates_split(gates_pre, x, h), packed[1], bias=conv_hh.bias.detach().float())
            return hipops.convgru_update_split(gates_pre, cand_pre, h, prelu_w, x_next)
        if isinstance(xh, hipops.SplitAct):
            raise RuntimeError('ConvGRU: a split-format input reached a step that runs library convolutions')
        if packed is not None:
            gates_pre = hipops.conv2d_mfma_sx(hipops.act_split(xh), packed[0], bias=conv_ih.bias.detach().float())
            xrh = hipops.convgru_gates(gates_pre, x, h)
            cand_pre = hipops.conv2d_mfma_sx(hipops.act_split(xrh), packed[1], bias=conv_hh.bias.detach().float())
        else:
            gates_pre = F.conv2d(xh, conv_ih.weight, conv_ih.bias, padding=conv_ih.padding)
            xrh = hipops.convgru_gates(gates_pre, x, h)
            cand_pre = F.conv2d(xrh, conv_hh.weight, conv_hh.bias, padding=conv_hh.padding)
        return hipops.convgru_update(gates_pre, cand_pre, h, prelu_w, x_next)

    def forward_single_frame(self, x, h):
        if self._fused(x):
            x, h = x.contiguous(), h.contiguous()
            h, _ = self._step_fused(torch.cat([x, h], dim=1), x, h, None)
            return h, h
        r, z = self.ih(torch.cat([x, h], dim=1)).split(self.channels, dim=1)
        c = self.hh(torch.cat([x, r * h], dim=1))
        h = (1 - z) * h + z * c
        return h, h

    def forward_time_series(self, x, h, seq2seq):
        outs = []
        if self._fused(x):
            frames = [xt.contiguous() for xt in x.unbind(dim=1)]
            h = h.contiguous()
            xh = torch.cat([frames[0], h], dim=1)
            for t, xt in enumerate(frames):
                h, xh = self._step_fused(xh, xt, h, frames[t + 1] if t + 1 < len(frames) else None)
                if seq2seq:
                    outs.append(h)
            return (torch.stack(outs, dim=1) if seq2seq else h), h
        for xt in x.unbind(dim=1):
            ot, h = self.forward_single_frame(xt, h)
            if seq2seq:
                outs.append(ot)
        return (torch.stack(outs, dim=1) if seq2seq else ot), h

    def forward(self, x, h, seq2seq=False):
        if h is None:
            h = torch.zeros((x.size(0), x.size(-3), x.size(-2), x.size(-1)), device=x.device, dtype=x.dtype)
        if x.ndim == 5:
            return self.forward_time_series(x, h, seq2seq)
        return self.forward_single_frame(x, h)


class DoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, use_instnorm=False):
        super().__init__()
        self.double_conv = nn.Sequential(
            nn.InstanceNorm2d(in_channels) if use_instnorm else nn.BatchNorm2d(in_channels),
            Conv2d(in_channels, out_channels, kernel_size=3, padding=1), nn.PReLU(out_channels),
            Conv2d(out_channels, out_channels, kernel_size=3, padding=1), nn.PReLU(out_channels), nn.PReLU(out_channels))

    def forward(self, x):
        from . import trunk_hip
        if HIP_GRU_CONVS and trunk_hip.double_conv_supported(self, x):
            return trunk_hip.double_conv_forward(self, x)
        return self.double_conv(x)


class Up(nn.Module):
    def __init__(self, in_channels, out_channels, upscale_factor=2):
        super().__init__()
        self.up = nn.PixelShuffle(upscale_factor=upscale_factor)
        self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        return self.conv(torch.cat([x2, self.up(x1)], dim=1))


class recurrent_Up(nn.Module):
    def __init__(self, in_channels, out_channels, upscale_factor=2):
        super().__init__()
        self.up = nn.PixelShuffle(upscale_factor=upscale_factor)
        self.conv = DoubleConv(in_channels, out_channels, use_instnorm=False)
        self.conv_gru = ConvGRU(out_channels, out_act_prelu=False)

    def forward(self, x1, x2, T, r=None, seq2seq=False):
        x = self.conv(torch.cat([x2, self.up(x1)], dim=1))
        return self.conv_gru(x.unflatten(0, (-1, T)), r, seq2seq)     # [B*T,C,H,W] -> [B,C,H,W]


class _UNetBase(Module):
    """Trunk + the four up-stages; subclasses add their heads."""

    def _build(self, inp_ch, res, use_gru):
        self.res = res
        self.use_gru = use_gru
        self.face_pool = None if res is None else torch.nn.AdaptiveAvgPool2d((res, res))
        self.input_layer, self.body = irse50_trunk(inp_ch)
        stage = recurrent_Up if use_gru else Up
        self.up1 = stage(1024, 512, upscale_factor=1)
        self.up2 = stage(384, 384)
        self.up3 = stage(224, 256)
        self.up4 = stage(128, 96)

    def _encode(self, x):
        if self.face_pool is not None and x.shape[-1] != self.res:
            x = face_pool_to(self.face_pool, x)
        x, (c0, c1, c2, c3) = run_trunk(self.body, self.input_layer(x), (2, 6, 20, 21))
        return x, c0, c1, c2, c3

    def gru_state_shapes(self, feats):
        """Shapes of the four ConvGRU states from the trunk features [x, c0, c1, c2, c3]: stage k (up1 .. up4) keeps one state of its
        cell's channel count at the resolution of the skip feature it concatenates (c3, c2, c1, c0) -- what a rank that does not own
        the chain needs to receive them (inversion_parallel)."""
        assert self.use_gru
        skips = (feats[4], feats[3], feats[2], feats[1])
        return [(1, up.conv_gru.channels) + tuple(skip.shape[-2:]) for up, skip in zip((self.up1, self.up2, self.up3, self.up4), skips)]

    def _decode(self, feats, T, r_list):
        """Yields the four decoder activations (16, 32, 64, 128^2); fills r_list in place when recurrent."""
        x, c0, c1, c2, c3 = feats
        t = None
        for k, (up, skip) in enumerate(zip((self.up1, self.up2, self.up3, self.up4), (c3, c2, c1, c0))):
            src = x if k == 0 else t
            if self.use_gru:
                if k > 0:   # the GRU collapsed T: broadcast its output back over the frames of the group
                    src = src.unsqueeze(1).expand(-1, T, -1, -1, -1).flatten(0, 1)
                t, r_list[k] = up(src, skip, T, r_list[k])
            else:
                t = up(src, skip)
            yield t


class TriPlanefeat_Encoder(_UNetBase):
    def __init__(self, inp_ch, seq2seq=False, res=None, use_gru=False):
        super().__init__()
        self.seq2seq = seq2seq
        self._build(inp_ch, res, use_gru)
        self.outconv0 = Conv2d(384, 32, kernel_size=1, padding=0)
        self.outconv1 = Conv2d(384, 512, kernel_size=1, padding=0)
        self.outconv2 = Conv2d(256, 512, kernel_size=1, padding=0)
        self.outconv3 = Conv2d(96, 256, kernel_size=1, padding=0)

    def forward_onlyEncoder(self, x):
        assert x.dim() == 5
        return list(self._encode(x.flatten(0, 1)))

    def _heads(self, feats, T, r_list):
        _, a32, a64, a128 = list(self._decode(feats, T, r_list))
        return [self.outconv0(a32), self.outconv1(a32), self.outconv2(a64), self.outconv3(a128)]

    def forward_onlyDecoder(self, T, cond_list, r_list=None):
        if self.use_gru:
            r_list = [None] * 4 if r_list is None else r_list
            return self._heads(cond_list, T, r_list), r_list
        return self._heads(cond_list, T, None)

    def forward(self, x, r_list=None, return_list=True):
        T = 1
        if x.dim() == 5:
            T = x.shape[1]
            x = x.flatten(0, 1)
        feats = self._encode(x)
        if self.use_gru:
            r_list = [None] * 4 if r_list is None else r_list
            return self._heads(feats, T, r_list), r_list
        return self._heads(feats, T, None)


class TriPlaneSFTfeat_Encoder(_UNetBase):
    def __init__(self, inp_ch, sft_half=True, res=None, use_gru=False):
        super().__init__()
        self.sft_half = sft_half
        self._build(inp_ch, res, use_gru)
        self.head = nn.PixelShuffle(upscale_factor=2)
        self.final_head = nn.Sequential(Conv2d(24, 96, kernel_size=3, padding=1), nn.PReLU(96),
                                        Conv2d(96, 96, kernel_size=3, padding=1), nn.PReLU(96))
        self.block_resolutions = [2 ** i for i in range(int(np.log2(16)), int(np.log2(256)) + 1)]
        gen_channels = {res: min(32768 // res, 512) for res in self.block_resolutions}
        dec_channels = {16: 512, 32: 384, 64: 256, 128: 96, 256: 96}
        for res in self.block_resolutions:
            ch, out = dec_channels[res], gen_channels[res] // 2 if sft_half else gen_channels[res]
            for kind in ('scale', 'shift'):
                setattr(self, f'condition_{kind}{res}', nn.Sequential(Conv2d(ch, ch, 3, 1, 1), nn.LeakyReLU(0.2, True),
                                                                     Conv2d(ch, out, 3, 1, 1)))

    def _sft(self, res, t):
        from . import trunk_hip
        scale, shift = getattr(self, f'condition_scale{res}'), getattr(self, f'condition_shift{res}')
        if HIP_GRU_CONVS and trunk_hip.conv_lrelu_conv_supported(scale, t) and trunk_hip.conv_lrelu_conv_supported(shift, t):
            from ... import hipops
            ts = hipops.act_split(t.contiguous())
            return torch.stack([trunk_hip.conv_lrelu_conv_forward(scale, ts), trunk_hip.conv_lrelu_conv_forward(shift, ts)])
        return torch.stack([scale(t), shift(t)])

    def forward_onlyEncoder(self, x):
        """The IR-SE50 trunk alone (eval-mode BatchNorm: every frame on its own), as TriPlanefeat_Encoder.forward_onlyEncoder: the
        half of this UNet that inversion_parallel shards by frame."""
        assert x.dim() == 5
        return list(self._encode(x.flatten(0, 1)))

    def forward_onlyDecoder(self, T, feats, r_list=None):
        """The recurrent decoder + CS-SFT heads on trunk features of T frames (see forward)."""
        if self.use_gru and r_list is None:
            r_list = [None] * 4
        out, t = {}, None
        for res, t in zip((16, 32, 64, 128), self._decode(feats, T, r_list)):
            out[res] = self._sft(res, t)
        out[256] = self._sft(256, self.final_head(self.head(t)))
        return (out, r_list) if self.use_gru else out

    def forward(self, x, r_list=None):
        T = 1
        if x.dim() == 5:
            T = x.shape[1]
            x = x.flatten(0, 1)
        feats = self._encode(x)
        if self.use_gru and r_list is None:
            r_list = [None] * 4
        out, t = {}, None
        for res, t in zip((16, 32, 64, 128), self._decode(feats, T, r_list)):
            out[res] = self._sft(res, t)
        out[256] = self._sft(256, self.final_head(self.head(t)))
        return (out, r_list) if self.use_gru else out
