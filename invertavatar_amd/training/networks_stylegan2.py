"""StyleGAN2 generator layers on the MI355X backend.

Public names, constructor arguments, parameter/buffer names and forward signatures follow the
reference (training/networks_stylegan2.py: modulated_conv2d :34, FullyConnectedLayer :96,
Conv2dLayer :132, MappingNetwork :190, SynthesisLayer :276, ToRGBLayer :340, SynthesisBlock :365,
SynthesisNetwork :470, Generator :517) so that reference checkpoints load by name
(SURVEY.md C14).  What differs is the execution on device tensors:

  * a SynthesisLayer is ONE fused MFMA convolution (``ia_conv2d_mfma``: style-scaled input patch,
    demodulation, noise, bias, lrelu, gain and clamp in the epilogue); up-sampling layers are the
    stride-2 transposed MFMA convolution followed by ``ia_upfirdn2d_bias_act`` (FIR + the same tail);
  * a ToRGBLayer is the 1x1 MFMA convolution with the bias and the up-sampled skip image added in
    its epilogue;
  * weights are repacked once per layer into the tap-major layout the kernel reads.

CPU tensors take the plain-torch route through ``conv2d_resample``/``bias_act`` exactly as the
reference does.  The discriminator classes of the reference file are training-only and out of scope.
"""
import math

import os

import numpy as np
import torch

from ..torch_utils import misc, persistence
from ..torch_utils.ops import bias_act, conv2d_resample, fma, upfirdn2d
from .. import _runtime, hipops

# Blocks built with use_fp16 (the reference's fp16 blocks: SR head with sr_num_fp16_res > 0) keep their activations in
# fp32 on this backend; FP16_BLOCKS_COMPUTE_FP32 = False additionally runs their 3x3 convolutions with fp16 operands and
# fp32 accumulation on the fp16 MFMA (ia_conv2d_mfma_h) -- the arithmetic precision of the reference's fp16 path without
# its fp16 storage rounding.  True (default) computes them entirely in fp32, a superset of the reference's precision.
FP16_BLOCKS_COMPUTE_FP32 = True

# fp32 layers: where the shape allows (hipops.conv_h_supported), the 3x3 convolutions form their fp32 products from hi/lo
# fp16 pairs on the fp16 MFMA (ia_conv2d_mfma_s): as accurate against an fp64 convolution as the fp32 MFMA form
# (tests/test_conv_gpu.py) and twice as fast.  False keeps every layer on v_mfma_f32_32x32x2_f32.
FUSED_TORGB = True               # a block whose x nobody reads (last SR block): ToRGB evaluated in conv1's epilogue (ia_conv2d_mfma_sx_rgb)
HIP_FC_LINEAR = True             # FullyConnectedLayer (linear activation, with bias) on a device matrix through ia_tokens_split + ia_linear_sx
STREAMING_TORGB = True           # ToRGB layers through ia_conv1x1 (one streaming launch) instead of the tiled ia_conv2d_mfma form
# r03 - r05: ia_torgb re-read the activations once per block of 32 output channels, and past 65 536 pixels x channel blocks (the 96-channel
# ToRGB of the static backbone at 256^2) ia_conv1x1 + ia_upfirdn2d shared the machine better (346.7 vs 344.2 frames/s).  Since r06 the large
# images read their activations once (torgb_wide_kernel) and the limit costs 0.6 % (same-box 413.8 -> 416.3 without it): no limit.
TORGB_MAX_WORK = 1 << 30
# Up-sampling layers with at most this many input channels run as one stride-1 launch on the weight composed with the resample filter
# (ia_upconv2d_fir_sx: 4x the products, no (2H+1)^2 fp32 image, no FIR launch): the 32 -> 256 @128^2 layer of the SR head
COMPOSED_UPFIR = True
COMPOSED_UPFIR_MAX_IN = 32
# Up-sampling layers from UPCONV_ROWS_MIN_RES^2 inputs: the transposed convolution per output row phase on the stride-1 tile
# (ia_upconv2d_rows_sx, csrc/conv_up.hip) instead of the four-phase tile of ia_conv2d_mfma_sx.
UPCONV_ROWS = True
UPCONV_ROWS_MIN_RES = 32      # (same-box frame A/B, 5 rounds: from 64^2 364.5, from 32^2 367.6 frames/s; at 16^2 the four-phase tile is twice as fast alone)
FUSED_TORGB_SKIP = True          # ... and, where ia_torgb covers the shape, with the skip image's up-sampling + add in the same launch
SPLIT_FP16_PRODUCTS = True

# ... and where the INPUT can be had as fp16 hi/lo planes (hipops.SplitAct: written by the producing layer's epilogue, or by
# ia_act_split), those layers run on ia_conv2d_mfma_sx: same arithmetic, same bits, operands DMA'd into LDS.  False keeps the
# register-staged kernel (ia_conv2d_mfma_s) everywhere.
USE_SPLIT_DMA = True


@misc.profiled_function
def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


class _PackedWeights(_runtime.DeviceCache):
    """Per-layer cache of the kernel-side weight layouts, rebuilt when the parameter changes."""

    def __init__(self):
        self.key = None
        self.wk = None
        self.wsq = None
        self.wk_h = None

    def get(self, weight, scale=1.0):
        key = (weight.data_ptr(), weight._version, weight.device, weight.dtype, scale)
        if key != self.key:
            w32 = weight.detach().float() * scale if scale != 1.0 else weight.detach().float()
            self.wk = hipops.pack_conv_weight(w32)
            self.wsq = hipops.weight_sq_sum(w32)
            self.wk_h = None
            self.key = key
        return self.wk, self.wsq

    def get_split(self, weight):
        """hi/lo fp16 packing for ia_conv2d_mfma_s (made on first use)."""
        self.get(weight)
        if getattr(self, 'wk_s', None) is None or self.wk_s_key != self.key:
            self.wk_s = hipops.pack_conv_weight_split(weight.detach().float())
            self.wk_s_key = self.key
        return self.wk_s

    def get_upfir(self, weight, resample_filter, half):
        """Packing of the weight composed with the resample filter (hipops.compose_upfir_weight) for ia_upconv2d_fir_sx."""
        self.get(weight)
        key = (self.key, resample_filter.data_ptr(), resample_filter._version, bool(half))
        if getattr(self, 'wk_uf_key', None) != key:
            wc = hipops.compose_upfir_weight(weight, resample_filter)
            self.wk_uf = hipops.pack_conv_weight_h(wc) if half else hipops.pack_conv_weight_split(wc)
            self.wk_uf_key = key
        return self.wk_uf

    def get_half(self, weight):
        """fp16 packing for ia_conv2d_mfma_h (made on first use)."""
        self.get(weight)
        if self.wk_h is None:
            self.wk_h = hipops.pack_conv_weight_h(weight.detach().float())
        return self.wk_h


def _on_device(x):
    return x.device.type == 'cuda'


def _needs_autograd(*tensors):
    """The fused HIP stages are forward-only; anything that must be differentiated takes the torch route."""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _hip_conv_ok(x, weight, up, down):
    kh, kw = weight.shape[2:]
    return (_on_device(x) and x.dtype == torch.float32 and down == 1 and kh == kw and
            ((up == 1 and kh in (1, 3)) or (up == 2 and kh == 3)))


@misc.profiled_function
def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True):
    """Modulated convolution, reference signature (training/networks_stylegan2.py:34-48).

    Device fp32 tensors with the generator's layer shapes run on the MFMA kernel; everything else
    (CPU, exotic shapes) follows the reference's two torch formulations."""
    batch = x.shape[0]
    out_ch, in_ch, kh, kw = weight.shape
    misc.assert_shape(weight, [out_ch, in_ch, kh, kw])
    misc.assert_shape(x, [batch, in_ch, None, None])
    misc.assert_shape(styles, [batch, in_ch])

    noise_ok = noise is None or noise.numel() == (x.shape[2] * up) * (x.shape[3] * up)   # a per-pixel table the kernel can index
    if (_hip_conv_ok(x, weight, up, down) and padding == kh // 2 and flip_weight == (up == 1) and noise_ok
            and (up == 1 or resample_filter is not None) and not _needs_autograd(x, weight, styles, noise)):
        w32 = weight.float()
        wk = hipops.pack_conv_weight(w32)
        styles = styles.float().contiguous()
        demod = hipops.modconv_demod(styles, hipops.weight_sq_sum(w32)) if demodulate else None
        x = x.contiguous()
        nz = None if noise is None else noise.reshape(-1).float().contiguous()
        if up == 1:
            return hipops.conv2d_mfma(x, wk, styles, demod, noise=nz, ksize=kh)
        t = hipops.conv2d_mfma(x, wk, styles, demod, ksize=3, transposed=True)
        return hipops.upfirdn2d_bias_act(t, resample_filter, noise=nz, up=1, pad0=(1, 1),
                                         out_hw=(x.shape[2] * 2, x.shape[3] * 2), fir_gain=4.0)

    # torch formulations (CPU tensors, autograd, and shapes without a kernel)
    if x.dtype == torch.float16 and demodulate:
        # fp16 head-room: both factors are brought to unit max-norm first; the demodulation cancels the scales again
        weight = weight * (1 / np.sqrt(in_ch * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)
    resample = dict(f=resample_filter, up=up, down=down, padding=padding, flip_weight=flip_weight)
    if fused_modconv:
        return _modconv_sample_weights(x, weight, styles, noise, demodulate, resample)
    return _modconv_scaled_input(x, weight, styles, noise, demodulate, resample)


def _modulated_weights(weight, styles):
    """[B, O, I, kh, kw]: the shared weight with every input channel scaled by the sample's style."""
    return weight[None] * styles[:, None, :, None, None]


def _demod_coefficients(weight, styles):
    """[B, O]: 1 / l2-norm of each sample's modulated filter (training/networks_stylegan2.py:63-64)."""
    return (_modulated_weights(weight, styles).square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt()


def _modconv_scaled_input(x, weight, styles, noise, demodulate, resample):
    """Styles applied to the activations, one shared weight, demodulation on the output (the reference's training-time
    route, :70-79)."""
    batch = x.shape[0]
    dtype = x.dtype
    y = conv2d_resample.conv2d_resample(x=x * styles.to(dtype).reshape(batch, -1, 1, 1), w=weight.to(dtype), **resample)
    d = _demod_coefficients(weight, styles).to(dtype).reshape(batch, -1, 1, 1) if demodulate else None
    if d is not None and noise is not None:
        return fma.fma(y, d, noise.to(dtype))
    if d is not None:
        return y * d
    return y if noise is None else y.add_(noise.to(dtype))


def _modconv_sample_weights(x, weight, styles, noise, demodulate, resample):
    """Per-sample weights (modulated, then demodulated) run as ONE grouped convolution with the batch folded into the
    channel axis (the reference's inference-time route, :81-91)."""
    batch, _, h, w_ = x.shape
    wb = _modulated_weights(weight, styles)
    if demodulate:
        wb = wb * _demod_coefficients(weight, styles).reshape(batch, -1, 1, 1, 1)
    y = conv2d_resample.conv2d_resample(x=x.reshape(1, -1, h, w_), w=wb.reshape(-1, *weight.shape[1:]).to(x.dtype), groups=batch,
                                        **resample)
    y = y.reshape(batch, -1, *y.shape[2:])
    return y if noise is None else y.add_(noise)


@persistence.persistent_class
class FullyConnectedLayer(torch.nn.Module):
    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier
        self._scaled = None   # (key, weight^T * gain, bias * gain): inference-time cache of the equalised-lr scaling
        self._split = None    # (key, fp16-pair split of weight * gain, bias * gain): the operand of ia_linear_sx

    def _scaled_params(self, dtype):
        key = (self.weight.data_ptr(), self.weight._version, None if self.bias is None else self.bias._version, dtype, self.weight.device)
        if self._scaled is None or self._scaled[0] != key:
            wt = (self.weight.detach().to(dtype) * self.weight_gain).t().contiguous()
            b = None if self.bias is None else (self.bias.detach().to(dtype) * self.bias_gain).unsqueeze(0).contiguous()
            self._scaled = (key, wt, b)
        return self._scaled[1], self._scaled[2]

    def forward(self, x):
        if not _needs_autograd(x, self.weight, self.bias):
            if (HIP_FC_LINEAR and self.activation == 'linear' and self.bias is not None and _on_device(x) and x.dtype == torch.float32 and x.dim() == 2
                    and self.in_features % 16 == 0):
                # the e4e heads' 512 x 512 layers on one row: the last library GEMM of the inversion flows goes through ia_linear_sx
                key = (self.weight.data_ptr(), self.weight._version, self.bias._version, self.weight.device)
                if getattr(self, '_split', None) is None or self._split[0] != key:
                    self._split = (key, hipops.pack_linear_weight_split(self.weight.detach().float() * self.weight_gain),
                                   (self.bias.detach().float() * self.bias_gain).contiguous())
                return hipops.linear_sx(hipops.tokens_split(x.contiguous()), self._split[1], self._split[2])
            wt, b = self._scaled_params(x.dtype)
            if self.activation == 'linear' and b is not None:
                return torch.addmm(b, x, wt)
            return bias_act.bias_act(x.matmul(wt), None if b is None else b.squeeze(0), act=self.activation)
        w = self.weight.to(x.dtype) * self.weight_gain
        b = self.bias
        if b is not None:
            b = b.to(x.dtype)
            if self.bias_gain != 1:
                b = b * self.bias_gain
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)

    def extra_repr(self):
        return f'in_features={self.in_features:d}, out_features={self.out_features:d}, activation={self.activation:s}'


@persistence.persistent_class
class Conv2dLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1,
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.activation = activation
        self.up = up
        self.down = down
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt)
        bias = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(bias) if bias is not None else None
        else:
            self.register_buffer('weight', weight)
            if bias is not None:
                self.register_buffer('bias', bias)
            else:
                self.bias = None

    def forward(self, x, gain=1):
        w = self.weight * self.weight_gain
        b = self.bias.to(x.dtype) if self.bias is not None else None
        x = conv2d_resample.conv2d_resample(x=x, w=w.to(x.dtype), f=self.resample_filter, up=self.up, down=self.down,
                                            padding=self.padding, flip_weight=(self.up == 1))
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return bias_act.bias_act(x, b, act=self.activation, gain=self.act_gain * gain, clamp=clamp)

    def extra_repr(self):
        return (f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, '
                f'activation={self.activation:s}, up={self.up}, down={self.down}')


@persistence.persistent_class
class MappingNetwork(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None,
                 activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.998):
        super().__init__()
        self.z_dim = z_dim
        self.c_dim = c_dim
        self.w_dim = w_dim
        self.num_ws = num_ws
        self.num_layers = num_layers
        self.w_avg_beta = w_avg_beta
        if embed_features is None:
            embed_features = w_dim
        if c_dim == 0:
            embed_features = 0
        if layer_features is None:
            layer_features = w_dim
        widths = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for idx in range(num_layers):
            setattr(self, f'fc{idx}', FullyConnectedLayer(widths[idx], widths[idx + 1], activation=activation,
                                                          lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def _embed_inputs(self, z, c):
        """Unit-second-moment latent, concatenated with the unit-second-moment embedding of the label (:233-243)."""
        parts = []
        if self.z_dim > 0:
            misc.assert_shape(z, [None, self.z_dim])
            parts.append(normalize_2nd_moment(z.to(torch.float32)))
        if self.c_dim > 0:
            misc.assert_shape(c, [None, self.c_dim])
            parts.append(normalize_2nd_moment(self.embed(c.to(torch.float32))))
        return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        w = self._embed_inputs(z, c)
        for idx in range(self.num_layers):
            w = getattr(self, f'fc{idx}')(w)
        if update_emas and self.w_avg_beta is not None:     # running mean of w (training only)
            self.w_avg.copy_(w.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:                         # one copy of w per synthesis layer
            w = w.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:                             # truncation trick: pull (the first `cutoff` rows) towards w_avg
            if self.w_avg_beta is None:
                raise AssertionError('truncation needs the w_avg buffer')
            if self.num_ws is None or truncation_cutoff is None:
                w = self.w_avg.lerp(w, truncation_psi)
            else:
                head = w[:, :truncation_cutoff]
                w[:, :truncation_cutoff] = self.w_avg.lerp(head, truncation_psi)
        return w

    def extra_repr(self):
        return f'z_dim={self.z_dim:d}, c_dim={self.c_dim:d}, w_dim={self.w_dim:d}, num_ws={self.num_ws:d}'


@persistence.persistent_class
class SynthesisLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True, activation='lrelu',
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.w_dim = w_dim
        self.resolution = resolution
        self.up = up
        self.use_noise = use_noise
        self.activation = activation
        self.conv_clamp = conv_clamp
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = bias_act.activation_funcs[activation].def_gain
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self._packed = _PackedWeights()
        self._pre = None   # (styles, demod) computed for this call by the owning network's StylePlan

    def _takes_split_input(self, in_res, noise_mode='const', half_ops=False):
        """True when this layer, fed a [*, in_channels, in_res, in_res] activation, runs on ia_conv2d_mfma_sx."""
        return ((SPLIT_FP16_PRODUCTS or half_ops) and USE_SPLIT_DMA and self.weight.shape[2] == 3 and self.in_channels % 8 == 0
                and self.out_channels % 8 == 0 and not (self.use_noise and noise_mode == 'random') and self.activation in hipops.ACT_ID
                and hipops.conv_sx_supported(self.in_channels, self.out_channels, in_res, in_res, 3, self.up == 2))

    def _consumer_styles(self, split_for, out_res, noise_mode, half_ops=False):
        """Styles of the layer that will consume this layer's result, if it can take it in split format (they were computed ahead
        of the forward pass by the network's _StyleBatcher and are parked on the consumer as `_pre`)."""
        if split_for is None or split_for._pre is None or split_for.in_channels != self.out_channels:
            return None
        return split_for._pre[0] if split_for._takes_split_input(out_res, noise_mode, half_ops) else None

    def _split_input(self, x, styles, planes):
        """This layer's input as the SplitAct ia_conv2d_mfma_sx reads (`planes` fp16 planes of x * styles).  A SplitAct that was
        made for this layer with that plane count is taken as it is; a SplitAct made for this layer with ANOTHER plane count
        already carries the styles and is only re-split; an unscaled one is scaled here; one made for a different layer is a
        producer / consumer mismatch and refused (its values are multiplied by somebody else's styles)."""
        carried = x if isinstance(x, hipops.SplitAct) else getattr(x, '_ia_split', None)
        if carried is not None and carried.consumer is self and carried.planes == planes:
            return carried
        if isinstance(x, hipops.SplitAct):
            if x.consumer is self:
                return hipops.act_split(x.float().contiguous(), None, consumer=self, planes=planes)
            if x.consumer is not None:
                raise RuntimeError('a SplitAct scaled for another layer reached this layer (producer/consumer out of sync)')
        return hipops.act_split(x.float().contiguous(), styles.float().contiguous(), consumer=self, planes=planes)

    def _fused_device_forward(self, x, styles, noise_mode, act_gain, act_clamp, demod=None, half_ops=False, split_for=None, keep_f32=True,
                              next_half_ops=None):
        """conv + demod + noise + bias + lrelu + clamp on the MFMA path (one or two launches).

        `x`: fp32 tensor (optionally carrying `_ia_split`, its split copy made for this layer) or a hipops.SplitAct.
        `split_for`: the SynthesisLayer that consumes the result; when it can take split input the result is ALSO (keep_f32) or
        ONLY (not keep_f32) produced as a SplitAct multiplied by that layer's styles."""
        in_res = self.resolution // self.up
        const_noise = self.use_noise and noise_mode == 'const'
        res = self.resolution
        if self._takes_split_input(in_res, noise_mode, half_ops):
            # half_ops (a block in the reference's fp16 precision): ONE fp16 plane in and out -- fp16 operands, fp32 accumulation,
            # and the activation between the convolutions stored as 2 bytes per element
            planes = 1 if half_ops else 2
            wk = self._packed.get_half(self.weight) if half_ops else self._packed.get_split(self.weight)
            if demod is None:
                demod = hipops.modconv_demod(styles.float().contiguous(), self._packed.get(self.weight)[1])
            xs = self._split_input(x, styles, planes)
            nz = self.noise_const.reshape(-1) if const_noise else None
            ns = self.noise_strength.detach().float().reshape(1) if const_noise else None
            bias = self.bias.detach().float()
            next_half = half_ops if next_half_ops is None else next_half_ops
            sn = self._consumer_styles(split_for, res, noise_mode, next_half)
            out_planes = 1 if next_half else 2
            if self.up == 1:
                out = hipops.conv2d_mfma_sx(xs, wk, demod, nz, ns, bias, act=self.activation, gain=act_gain, clamp=act_clamp,
                                            want_f32=keep_f32 or sn is None, split_for=split_for if sn is not None else None, styles_next=sn,
                                            split_planes=out_planes)
            elif (COMPOSED_UPFIR and sn is not None and self.in_channels <= COMPOSED_UPFIR_MAX_IN and tuple(self.resample_filter.shape) == (4, 4)
                  and hipops.upconv_fir_supported(xs.shape[0], self.in_channels, self.out_channels, in_res, in_res)):
                # few input channels: transposed convolution + FIR + tail as ONE stride-1 launch on the composed weight
                out = hipops.upconv_fir_sx(xs, self._packed.get_upfir(self.weight, self.resample_filter, half_ops), demod, nz, ns, bias, styles_next=sn,
                                           act=self.activation, gain=act_gain, clamp=act_clamp, want_f32=keep_f32, split_for=split_for,
                                           split_planes=out_planes)
            else:
                if UPCONV_ROWS and planes == 2 and in_res >= UPCONV_ROWS_MIN_RES and hipops.upconv_rows_supported(
                        xs.shape[0], self.in_channels, self.out_channels, in_res, in_res):
                    t = hipops.upconv2d_rows_sx(xs, wk, demod)      # per output row phase on the stride-1 tile (csrc/conv_up.hip)
                else:
                    t = hipops.conv2d_mfma_sx(xs, wk, demod, transposed=True)
                if sn is None:
                    return hipops.upfirdn2d_bias_act(t, self.resample_filter, nz, ns, bias, up=1, pad0=(1, 1), out_hw=(res, res),
                                                     fir_gain=4.0, act=self.activation, act_gain=act_gain, clamp=act_clamp)
                out = hipops.fir_tail_split(t, self.resample_filter, nz, ns, bias, styles_next=sn, out_hw=(res, res), pad0=(1, 1), fir_gain=4.0,
                                            act=self.activation, act_gain=act_gain, clamp=act_clamp, want_f32=keep_f32, split_for=split_for,
                                            planes=out_planes)
            if isinstance(out, tuple):     # fp32 result with its split copy riding along for the consumer
                out[0]._ia_split = out[1]
                return out[0]
            return out
        if isinstance(x, hipops.SplitAct):
            if x.consumer is not None:     # (scaled by some layer's styles: the fp32 values cannot be recovered exactly)
                raise RuntimeError('a SplitAct reached a layer that cannot consume it (producer/consumer eligibility out of sync)')
            x = x.float()
        wk, wsq = self._packed.get(self.weight)
        if (half_ops or SPLIT_FP16_PRODUCTS) and hipops.conv_h_supported(self.in_channels, self.out_channels, x.shape[2], x.shape[3], 3,
                                                                        self.up == 2):
            wk = self._packed.get_half(self.weight) if half_ops else self._packed.get_split(self.weight)
        styles = styles.float().contiguous()
        if demod is None:
            demod = hipops.modconv_demod(styles, wsq)
        x = x.float().contiguous()
        nz = self.noise_const.reshape(-1) if const_noise else None
        ns = self.noise_strength.detach().float().reshape(1) if const_noise else None
        bias = self.bias.detach().float()
        if self.use_noise and noise_mode == 'random':  # per-sample noise: keep it outside the fused tail
            rnd_noise = torch.randn([x.shape[0], 1, res, res], device=x.device) * self.noise_strength
            if self.up == 1:
                y = hipops.conv2d_mfma(x, wk, styles, demod, ksize=3)
            else:
                t = hipops.conv2d_mfma(x, wk, styles, demod, ksize=3, transposed=True)
                y = hipops.upfirdn2d_bias_act(t, self.resample_filter, up=1, pad0=(1, 1), out_hw=(res, res), fir_gain=4.0)
            return bias_act.bias_act(y + rnd_noise, bias, act=self.activation, gain=act_gain, clamp=act_clamp)
        if self.up == 1:
            return hipops.conv2d_mfma(x, wk, styles, demod, nz, ns, bias, ksize=3, act=self.activation, gain=act_gain,
                                      clamp=act_clamp)
        t = hipops.conv2d_mfma(x, wk, styles, demod, ksize=3, transposed=True)
        next_half = half_ops if next_half_ops is None else next_half_ops
        sn = self._consumer_styles(split_for, res, noise_mode, next_half)
        if sn is not None and self.out_channels % 8 == 0:      # (the consumer is on the split path although this layer is not)
            out = hipops.fir_tail_split(t, self.resample_filter, nz, ns, bias, styles_next=sn, out_hw=(res, res), pad0=(1, 1), fir_gain=4.0,
                                        act=self.activation, act_gain=act_gain, clamp=act_clamp, want_f32=keep_f32, split_for=split_for,
                                        planes=1 if next_half else 2)
            if isinstance(out, tuple):
                out[0]._ia_split = out[1]
                return out[0]
            return out
        return hipops.upfirdn2d_bias_act(t, self.resample_filter, nz, ns, bias, up=1, pad0=(1, 1), out_hw=(res, res),
                                         fir_gain=4.0, act=self.activation, act_gain=act_gain, clamp=act_clamp)

    def forward_with_torgb(self, x, w, torgb, w_rgb, skip, resample_filter, noise_mode='random', gain=1, half_ops=False,
                           skip_upsampled=None, **_unused):
        """This layer AND the ToRGB layer that is the only reader of its result, in one launch (ia_conv2d_mfma_sx_rgb): returns the
        image `upsample2d(skip) + torgb(conv(x))`, or None when the pair is not eligible (the caller then runs the two layers)."""
        res = self.resolution
        if not (FUSED_TORGB and self.up == 1 and _on_device(x) and self.activation in hipops.ACT_ID and self.weight.shape[2] == 3
                and torgb.weight.shape[2] == 1 and torgb.out_channels <= 3 and torgb.in_channels == self.out_channels
                and noise_mode != 'random' and not _needs_autograd(x, w, w_rgb, self.weight, self.bias, torgb.weight, torgb.bias, skip)
                and self._takes_split_input(res, noise_mode, half_ops)
                and hipops.conv_sx_rgb_supported(x.shape[0], self.in_channels, self.out_channels, res, res)):
            return None
        pre, self._pre = self._pre, None
        styles, demod = pre if pre is not None else (self.affine(w), None)
        pre_rgb, torgb._pre = torgb._pre, None
        rgb_styles = pre_rgb[0] if pre_rgb is not None else torgb.affine(w_rgb).float().contiguous()
        planes = 1 if half_ops else 2
        wk = self._packed.get_half(self.weight) if half_ops else self._packed.get_split(self.weight)
        if demod is None:
            demod = hipops.modconv_demod(styles.float().contiguous(), self._packed.get(self.weight)[1])
        xs = self._split_input(x, styles, planes)
        const_noise = self.use_noise and noise_mode == 'const'
        nz = self.noise_const.reshape(-1) if const_noise else None
        ns = self.noise_strength.detach().float().reshape(1) if const_noise else None
        rgb_wk, _ = torgb._packed.get(torgb.weight, scale=torgb.weight_gain)
        if skip is not None and skip_upsampled is not None:      # (made by the caller beside the previous layers, see _TwoBlockHead)
            residual = skip_upsampled
        else:
            residual = None if skip is None else upfirdn2d.upsample2d(skip, resample_filter).float().contiguous()
        _, _, img = hipops.conv2d_mfma_sx_rgb(xs, wk, rgb_wk, rgb_styles, torgb.bias.detach().float(), residual, torgb.conv_clamp, demod, nz, ns,
                                              self.bias.detach().float(), act=self.activation, gain=self.act_gain * gain,
                                              clamp=self.conv_clamp * gain if self.conv_clamp is not None else None)
        return img

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1, half_ops=False, split_for=None, keep_f32=True, next_half_ops=None):
        assert noise_mode in ['random', 'const', 'none']
        in_res = self.resolution // self.up
        if isinstance(x, hipops.SplitAct):
            assert tuple(x.shape[1:]) == (self.in_channels, in_res, in_res), (x.shape, self.in_channels, in_res)
        else:
            misc.assert_shape(x, [None, self.in_channels, in_res, in_res])
        pre, self._pre = self._pre, None
        act_gain = self.act_gain * gain
        act_clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        if (_on_device(x) and self.activation in hipops.ACT_ID and self.weight.shape[2] == 3 and self.up in (1, 2)
                and not _needs_autograd(x, w, self.weight, self.bias)):
            styles, demod = pre if pre is not None else (self.affine(w), None)
            return self._fused_device_forward(x, styles, noise_mode, act_gain, act_clamp, demod, half_ops, split_for, keep_f32, next_half_ops)
        if isinstance(x, hipops.SplitAct):
            raise RuntimeError('a SplitAct reached the torch route of a SynthesisLayer')
        styles = self.affine(w)
        noise = None
        if self.use_noise and noise_mode == 'random':
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device) * self.noise_strength
        if self.use_noise and noise_mode == 'const':
            noise = self.noise_const * self.noise_strength
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, noise=noise, up=self.up, padding=self.padding,
                             resample_filter=self.resample_filter, flip_weight=(self.up == 1), fused_modconv=fused_modconv)
        return bias_act.bias_act(x, self.bias.to(x.dtype), act=self.activation, gain=act_gain, clamp=act_clamp)

    def extra_repr(self):
        return (f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, w_dim={self.w_dim:d}, '
                f'resolution={self.resolution:d}, up={self.up}, activation={self.activation:s}')


@persistence.persistent_class
class ToRGBLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.w_dim = w_dim
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self._packed = _PackedWeights()
        self._pre = None

    def forward(self, x, w, fused_modconv=True, residual=None, skip=None, resample_filter=None):
        """`residual` (fp32, output-shaped) is added after the clamp: the skip-image add of SynthesisBlock folded into this layer's
        epilogue on the device path.  `skip` (+ `resample_filter`): the PREVIOUS block's image, up-sampled 2x here and added as the residual
        (a single streaming launch for conv + up-sampling + add was measured in r02 and dropped: load-latency bound at B = 1, 42-140 us
        where the MFMA form + FIR take 30-80)."""
        if _on_device(x) and self.weight.shape[2] == 1 and not _needs_autograd(x, w, self.weight, self.bias, residual, skip):
            # weight_gain is folded into the packed weight instead of scaling the styles on every call
            wk, _ = self._packed.get(self.weight, scale=self.weight_gain)
            pre, self._pre = self._pre, None
            styles = pre[0] if pre is not None else self.affine(w).float().contiguous()
            x = x.float().contiguous()
            if (FUSED_TORGB_SKIP and STREAMING_TORGB and residual is None
                    and x.shape[2] * x.shape[3] * ((wk.shape[-1] + 31) // 32) <= TORGB_MAX_WORK
                    and hipops.torgb_supported(x.shape[1], wk.shape[-1], x.shape[2], x.shape[3], skip is not None)
                    and (skip is None or (resample_filter is not None and tuple(resample_filter.shape) == (4, 4) and skip.dtype == torch.float32
                                          and skip.shape[2] * 2 == x.shape[2] and skip.shape[3] * 2 == x.shape[3]))):
                # conv + bias + clamp + upsample2d(skip) + add in one launch (ia_torgb)
                return hipops.torgb(x, wk, styles, bias=self.bias.detach().float(), skip=None if skip is None else skip.contiguous(),
                                    skip_filter=None if skip is None else resample_filter.float().contiguous(), clamp=self.conv_clamp)
            if skip is not None:
                residual = upfirdn2d.upsample2d(skip, resample_filter)
            res = None if residual is None else residual.float().contiguous()
            if STREAMING_TORGB and hipops.conv1x1_supported(x.shape[1], wk.shape[-1], x.shape[2], x.shape[3]):
                return hipops.conv1x1(x, wk, styles, bias=self.bias.detach().float(), residual=res, clamp=self.conv_clamp)
            return hipops.conv2d_mfma(x, wk, styles, None,
                                      bias=self.bias.detach().float(), residual=res, ksize=1, act='linear', clamp=self.conv_clamp)
        if skip is not None:
            residual = upfirdn2d.upsample2d(skip, resample_filter)
        styles = self.affine(w) * self.weight_gain
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        x = bias_act.bias_act(x, self.bias.to(x.dtype), clamp=self.conv_clamp)
        if residual is not None:
            x = residual + x.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        return x

    def extra_repr(self):
        return f'in_channels={self.in_channels:d}, out_channels={self.out_channels:d}, w_dim={self.w_dim:d}'


@persistence.persistent_class
class SynthesisBlock(torch.nn.Module):
    """One resolution of the synthesis network ('skip' architecture on the generator path).

    `condition` = (scale, shift) applies the CS-SFT modulation of the inversion encoder to the upper
    half of the channels between the two convolutions
    (training_avatar_texture/networks_stylegan2_new.py:448-452); the stock block never passes it."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture='skip',
                 resample_filter=[1, 3, 3, 1], conv_clamp=256, use_fp16=False, fp16_channels_last=False,
                 fused_modconv_default=True, **layer_kwargs):
        assert architecture in ['orig', 'skip', 'resnet']
        super().__init__()
        self.in_channels = in_channels
        self.w_dim = w_dim
        self.resolution = resolution
        self.img_channels = img_channels
        self.is_last = is_last
        self.architecture = architecture
        self.use_fp16 = use_fp16
        self.channels_last = (use_fp16 and fp16_channels_last)
        self.fused_modconv_default = fused_modconv_default
        self.register_buffer('resample_filter', upfirdn2d.setup_filter(resample_filter))
        self.num_conv = 0
        self.num_torgb = 0
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        if in_channels != 0:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp,
                                        channels_last=self.channels_last, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp,
                                    channels_last=self.channels_last, **layer_kwargs)
        self.num_conv += 1
        if is_last or architecture == 'skip':
            self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp,
                                    channels_last=self.channels_last)
            self.num_torgb += 1
        if in_channels != 0 and architecture == 'resnet':
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2,
                                    resample_filter=resample_filter, channels_last=self.channels_last)

    def _half_ops(self, device, force_fp32=False):
        """True when this block's 3x3 convolutions run with fp16 operands (one-plane SplitAct in and out)."""
        return bool(self.use_fp16 and not force_fp32 and device.type == 'cuda' and not FP16_BLOCKS_COMPUTE_FP32)

    def forward(self, x, img, ws, condition=None, force_fp32=False, fused_modconv=None, update_emas=False, _next_conv=None, _next_half=None,
                _x_unused=False, _img_stream=None, _img_wait=None, _skip_upsampled=None, **layer_kwargs):
        """`_next_conv`: the layer that consumes this block's x (the next block's conv0), given by the owning network on the device
        inference path so that conv1 can emit its result in the format that layer reads (hipops.SplitAct).
        `_x_unused`: the caller drops the returned x (last block of a head): conv1 may then evaluate ToRGB in its epilogue and x comes
        back as None."""
        _ = update_emas
        misc.assert_shape(ws, [None, self.num_conv + self.num_torgb, self.w_dim])
        w_iter = iter(ws.unbind(dim=1))
        half_ops = self._half_ops(ws.device, force_fp32)
        force_fp32 = True      # storage stays fp32 on this backend (CPU: as the reference, :437)
        if half_ops:
            layer_kwargs = dict(layer_kwargs, half_ops=True)
        dtype = torch.float16 if self.use_fp16 and not force_fp32 else torch.float32
        fmt = torch.channels_last if self.channels_last and not force_fp32 else torch.contiguous_format
        if fused_modconv is None:
            fused_modconv = self.fused_modconv_default
        if fused_modconv == 'inference_only':
            fused_modconv = (not self.training)

        if self.in_channels == 0:
            x = self.const.to(dtype=dtype, memory_format=fmt).unsqueeze(0)
            if ws.shape[0] != 1:      # (one frame: conv1 reads the parameter in place -- no copy launch at the head of the network)
                x = x.repeat([ws.shape[0], 1, 1, 1])
        elif not isinstance(x, hipops.SplitAct):      # (a SplitAct was made for conv0 by its producer; conv0 checks its shape)
            misc.assert_shape(x, [None, self.in_channels, self.resolution // 2, self.resolution // 2])
            x = x.to(dtype=dtype, memory_format=fmt)

        if self.in_channels == 0:
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, split_for=_next_conv, **layer_kwargs)
        elif self.architecture == 'resnet':
            y = self.skip(x, gain=np.sqrt(0.5))
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, gain=np.sqrt(0.5), **layer_kwargs)
            x = y.add_(x)
        else:
            # (conv0's result has one consumer, conv1: unless the CS-SFT condition edits it in between it is produced in conv1's
            # input format only)
            chain = dict(split_for=self.conv1, keep_f32=False) if condition is None else {}
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **chain, **layer_kwargs)
            if condition is not None:
                half = int(x.size(1) // 2)
                x = torch.cat([x[:, :half], x[:, half:] * condition[0] + condition[1]], dim=1)
            fused_img = None
            if _img_wait is not None:      # the incoming skip image was made on a side stream (see _img_stream below): join it here
                torch.cuda.current_stream(x.device if torch.is_tensor(x) else ws.device).wait_stream(_img_wait)
            if _x_unused and condition is None and (self.is_last or self.architecture == 'skip'):
                w_conv1, w_rgb = next(w_iter), next(w_iter)
                fused_img = self.conv1.forward_with_torgb(x, w_conv1, self.torgb, w_rgb, img, self.resample_filter,
                                                          skip_upsampled=_skip_upsampled, **layer_kwargs)
                if fused_img is not None:
                    return None, fused_img.to(dtype=torch.float32, memory_format=torch.contiguous_format)
                w_iter = iter((w_conv1, w_rgb))
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, split_for=_next_conv, next_half_ops=_next_half, **layer_kwargs)

        if img is not None:
            misc.assert_shape(img, [None, self.img_channels, self.resolution // 2, self.resolution // 2])
        if self.is_last or self.architecture == 'skip':
            # img = upsample2d(img) + torgb(x): the add happens in the ToRGB epilogue
            if _img_stream is not None and torch.is_tensor(x) and x.is_cuda:
                # The skip image is not an input of the next block's convolutions: a caller whose machine is otherwise idle (the
                # SR head) lets it run beside them.  The caller keeps x alive until it has joined the stream (_img_wait).
                cur = torch.cuda.current_stream(x.device)
                _img_stream.wait_stream(cur)
                with torch.cuda.stream(_img_stream):
                    img = self.torgb(x, next(w_iter), fused_modconv=fused_modconv, skip=img, resample_filter=self.resample_filter)
                    img = img.to(dtype=torch.float32, memory_format=torch.contiguous_format)
            else:
                img = self.torgb(x, next(w_iter), fused_modconv=fused_modconv, skip=img, resample_filter=self.resample_filter)
                img = img.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        elif img is not None:
            img = upfirdn2d.upsample2d(img, self.resample_filter)

        assert x.dtype == dtype
        assert img is None or img.dtype == torch.float32
        return x, img

    def extra_repr(self):
        return f'resolution={self.resolution:d}, architecture={self.architecture:s}'


def _modulated_layers(block):
    """(layer, index into the block's ws, demodulate?) in the order SynthesisBlock.forward consumes ws."""
    out, k = [], 0
    if block.in_channels != 0:
        out.append((block.conv0, k, True)); k += 1
    out.append((block.conv1, k, True)); k += 1
    if block.is_last or block.architecture == 'skip':
        out.append((block.torgb, k, False))
    return out


class _StyleBatcher(_runtime.DeviceCache):
    """Computes the styles / demodulation coefficients of all layers of a list of blocks in two launches
    (``ia_styles_demod``) and parks them on the layers for the forward pass that follows."""

    def __init__(self):
        self.plan = None
        self.layers = None

    def prepare(self, blocks, ws_offsets, ws, enabled=True, fixed_w=None):
        """`fixed_w`: per block None or an index into ws that ALL of the block's layers read (the SR head: every layer sees the last w)."""
        if not (enabled and ws.is_cuda and not _needs_autograd(ws) and not torch.is_grad_enabled()):
            return False
        if self.plan is None or self.plan.device != ws.device:
            entries, self.layers = [], []
            for n, (block, off) in enumerate(zip(blocks, ws_offsets)):
                if block.architecture == 'resnet':
                    return False
                for layer, k, demod in _modulated_layers(block):
                    widx = off + k if fixed_w is None or fixed_w[n] is None else fixed_w[n]
                    entries.append(dict(affine=layer.affine, weight=layer.weight, widx=widx, demod=demod))
                    self.layers.append(layer)
            self.plan = hipops.StylePlan(entries, ws.device)
        for layer, pre in zip(self.layers, self.plan.run(ws)):
            layer._pre = pre
        return True


def _block_plan(img_resolution, channel_base, channel_max, num_fp16_res):
    log2 = int(np.log2(img_resolution))
    resolutions = [2 ** i for i in range(2, log2 + 1)]
    channels = {res: min(channel_base // res, channel_max) for res in resolutions}
    fp16_from = max(2 ** (log2 + 1 - num_fp16_res), 8)
    return log2, resolutions, channels, fp16_from


@persistence.persistent_class
class SynthesisNetwork(torch.nn.Module):
    """Stock StyleGAN2 synthesis network (training/networks_stylegan2.py:470-515)."""

    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=4, **block_kwargs):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.num_fp16_res = num_fp16_res
        self.img_resolution_log2, self.block_resolutions, channels, fp16_from = _block_plan(img_resolution, channel_base,
                                                                                          channel_max, num_fp16_res)
        self.num_ws = 0
        for res in self.block_resolutions:
            block = SynthesisBlock(channels[res // 2] if res > 4 else 0, channels[res], w_dim=w_dim, resolution=res,
                                   img_channels=img_channels, is_last=(res == img_resolution), use_fp16=(res >= fp16_from),
                                   **block_kwargs)
            self.num_ws += block.num_conv
            if res == img_resolution:
                self.num_ws += block.num_torgb
            setattr(self, f'b{res}', block)
        self._style_batcher = _StyleBatcher()

    def _style_blocks(self):
        """(blocks, index of each block's first w): what a style batcher needs to know about this network."""
        blocks = [getattr(self, f'b{res}') for res in self.block_resolutions]
        offs, idx = [], 0
        for blk in blocks:
            offs.append(idx)
            idx += blk.num_conv
        return blocks, offs

    def _prepare_styles(self, ws):
        """Batch every affine + demodulation of this network (device inference path only).  A caller that has batched SEVERAL
        networks' styles for this `ws` object already (triplane_v20: one launch pair per frame) marks them with `_styles_for`."""
        rt = _runtime.state(self)          # (runtime state lives outside the module: deepcopy / pickle see parameters only)
        marked, rt.styles_for = getattr(rt, 'styles_for', None), None
        if marked is not None and marked is ws:
            return
        self._style_batcher.prepare(*self._style_blocks(), ws.to(torch.float32))

    def _split_ws(self, ws):
        out, idx = [], 0
        with torch.autograd.profiler.record_function('split_ws'):
            misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
            ws = ws.to(torch.float32)
            for res in self.block_resolutions:
                block = getattr(self, f'b{res}')
                out.append(ws.narrow(1, idx, block.num_conv + block.num_torgb))
                idx += block.num_conv
        return out

    def forward(self, ws, **block_kwargs):
        x = img = None
        self._prepare_styles(ws)
        for res, cur_ws in zip(self.block_resolutions, self._split_ws(ws)):
            x, img = getattr(self, f'b{res}')(x, img, cur_ws, _next_conv=self._next_conv(res), _next_half=self._next_half(res, ws, block_kwargs),
                                              **block_kwargs)
        return img

    def _next_conv(self, res):
        """conv0 of the block after `res` (the consumer of this block's features), or None for the last block."""
        nxt = getattr(self, f'b{res * 2}', None)
        return getattr(nxt, 'conv0', None)

    def _next_half(self, res, ws, block_kwargs):
        """Does the block after `res` run its convolutions with fp16 operands?  (None: there is no next block.)"""
        nxt = getattr(self, f'b{res * 2}', None)
        return None if nxt is None else nxt._half_ops(ws.device, block_kwargs.get('force_fp32', False))

    def extra_repr(self):
        return (f'w_dim={self.w_dim:d}, num_ws={self.num_ws:d}, img_resolution={self.img_resolution:d}, '
                f'img_channels={self.img_channels:d}, num_fp16_res={self.num_fp16_res:d}')


@persistence.persistent_class
class Generator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim = z_dim
        self.c_dim = c_dim
        self.w_dim = w_dim
        self.img_resolution = img_resolution
        self.img_channels = img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
