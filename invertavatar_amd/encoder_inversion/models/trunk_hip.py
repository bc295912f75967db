"""IR-SE50 residual units of the inversion encoders on the MI355X convolution kernels.

Reference: encoder_inversion/models/helpers.py:102-124 (``bottleneck_IR_SE``): ``BatchNorm2d -> Conv2d 3x3 -> PReLU -> Conv2d 3x3
(stride s) -> BatchNorm2d -> SEModule`` plus the shortcut (``MaxPool2d(1, s)`` or ``Conv2d 1x1 (stride s) -> BatchNorm2d``).
Through ``torch.nn`` these are library convolutions (MIOpen fp32: 14-27 TFLOP/s on the trunk of a 4-frame group, r03 profile);
here the two 3x3 convolutions of a unit in EVAL mode run on ``ia_conv2d_mfma_sx`` (fp32 products from fp16 hi / lo pairs, the
arithmetic the generator's large layers use):

  * the first BatchNorm is an affine map per input channel: folded into the staging of conv1 (``ia_act_split`` with scale and
    shift: the zero padding then applies to the normalised tensor, as in the reference);
  * PReLU runs in conv1's epilogue (per-channel slopes), and conv1 hands its result to conv2 in split format, never as fp32;
  * the second BatchNorm is conv2's epilogue (scale = ``demod``, shift = ``bias`` per output channel);
  * a stride-2 conv2 is evaluated at stride 1 and sub-sampled (``y[..., ::2, ::2]`` is exactly the stride-2 result with padding 1):
    4 of the 48 convolutions of a trunk, +22 % FLOPs, no second kernel family.

The squeeze-and-excitation gate, the shortcut and the residual add stay ATen element-wise / tiny GEMM launches.  Units in TRAIN
mode (batch statistics: the e4e trunk under eval_seq.py's module modes), CPU tensors, autograd and layers below 32^2 (fewer points
than the 8-wave tile needs) take the unit's own ``torch.nn`` forward.
"""
import torch

from ... import _runtime, hipops


class _UnitPack(_runtime.DeviceCache):
    """Kernel-side parameters of one residual unit, rebuilt when a parameter / buffer changes."""

    def __init__(self):
        self.key = None

    def get(self, unit):
        bn1, conv1, prelu, conv2, bn2 = unit.res_layer[0], unit.res_layer[1], unit.res_layer[2], unit.res_layer[3], unit.res_layer[4]
        tensors = (bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, conv1.weight, prelu.weight, conv2.weight, bn2.weight, bn2.bias,
                   bn2.running_mean, bn2.running_var)
        key = tuple((t.data_ptr(), t._version) for t in tensors) + (conv1.weight.device,)
        if key != self.key:
            def affine(bn):
                a = (bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps))
                return a.contiguous(), (bn.bias.detach().float() - bn.running_mean.detach().float() * a).contiguous()
            self.a1, self.c1 = affine(bn1)
            self.a2, self.c2 = affine(bn2)
            self.w1 = hipops.pack_conv_weight_split(conv1.weight.detach().float())
            self.w2 = hipops.pack_conv_weight_split(conv2.weight.detach().float())
            self.slopes = prelu.weight.detach().float().contiguous()
            self.key = key
        return self


def unit_supported(unit, x):
    """True when the unit's two 3x3 convolutions can run on ia_conv2d_mfma_sx for this input."""
    if not (x.is_cuda and x.dtype == torch.float32 and not unit.training and not torch.is_grad_enabled()):
        return False
    conv1, conv2 = unit.res_layer[1], unit.res_layer[3]
    i, o = conv1.in_channels, conv1.out_channels
    h, w = x.shape[-2:]
    plain = all(c.padding == (1, 1) and c.dilation == (1, 1) and c.groups == 1 and c.padding_mode == 'zeros' for c in (conv1, conv2))
    return (plain and conv1.kernel_size == (3, 3) and conv2.kernel_size == (3, 3) and conv1.stride == (1, 1) and conv2.stride in ((1, 1), (2, 2))
            and conv1.bias is None and conv2.bias is None and i % 8 == 0 and o % 8 == 0 and o >= 64 and h * w >= 1024 and w <= 320
            and conv2.in_channels == o and conv2.out_channels == o)


def unit_forward(unit, x):
    """bottleneck_IR_SE.forward (helpers.py:121-124) with the residual branch's convolutions on the HIP kernels."""
    pack = _runtime.state(unit)
    if not hasattr(pack, 'trunk'):
        pack.trunk = _UnitPack()
    p = pack.trunk.get(unit)
    b, c, h, w = x.shape
    x = x.contiguous()
    stride = unit.res_layer[3].stride[0]
    xs = hipops.act_split(x, p.a1.unsqueeze(0).expand(b, -1).contiguous(), shift=p.c1.unsqueeze(0).expand(b, -1).contiguous())
    us = hipops.conv2d_mfma_sx(xs, p.w1, act='lrelu', prelu=p.slopes, want_f32=False, want_split=True)
    o = p.a2.numel()
    v = hipops.conv2d_mfma_sx(us, p.w2, demod=p.a2.unsqueeze(0).expand(b, -1).contiguous(), bias=p.c2, act='linear')
    if stride == 2:
        v = v[:, :, ::2, ::2]
    return se_tail(unit, v, x)


def se_tail(unit, v, x):
    """SEModule gate + shortcut + add of a residual unit (helpers.py:84-100, :121-124) in two launches (ia_se_gate); `v`: the residual
    branch in front of the gate (may be a strided view).  Units without a gate, CPU tensors and autograd take the ATen ops."""
    se = unit.res_layer[5] if len(unit.res_layer) > 5 else None
    short = unit.shortcut_layer
    if (se is not None and v.is_cuda and v.dtype == torch.float32 and not torch.is_grad_enabled() and se.fc1.out_channels <= 64
            and se.fc1.bias is None and se.fc2.bias is None):
        if isinstance(short, torch.nn.MaxPool2d) and short.kernel_size == 1:      # MaxPool2d(1, s) = the input sub-sampled
            s_ = short.stride if isinstance(short.stride, int) else short.stride[0]
            sc = x[:, :, ::s_, ::s_]
        else:
            sc = short(x)
        c = v.shape[1]
        return hipops.se_gate(v, sc, se.fc1.weight.detach().float().reshape(-1, c).contiguous(),
                              se.fc2.weight.detach().float().reshape(c, -1).contiguous())
    res = se(v) if se is not None else v
    return res + short(x)


# ------------------------------------------------------------------ plain 3x3 convolutions of the UNet decoders / heads
def conv_supported(conv, h, w):
    """A torch.nn.Conv2d(3x3, stride 1, padding 1) whose shape ia_conv2d_mfma_sx takes (8-wave tile: >= 1024 points, >= 64 outputs)."""
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros' and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0
            and conv.out_channels >= 64 and h * w >= 1024 and w <= 320)


def _device_path(x):
    return torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()


def packed_weight(conv):
    st = _runtime.state(conv)
    key = (conv.weight.data_ptr(), conv.weight._version, conv.weight.device)
    if getattr(st, 'wk_key', None) != key:
        st.wk = hipops.pack_conv_weight_split(conv.weight.detach().float())
        st.wk_key = key
    return st.wk


def conv3x3(xs, conv, slopes=None, alpha=None, want_split=False):
    """conv (+ bias) (+ leaky ReLU with scalar `alpha` or per-channel `slopes`) of a SplitAct; fp32 result, or a SplitAct for the
    next convolution (want_split)."""
    act = 'lrelu' if (slopes is not None or alpha is not None) else 'linear'
    bias = None if conv.bias is None else conv.bias.detach().float()
    return hipops.conv2d_mfma_sx(xs, packed_weight(conv), bias=bias, act=act, alpha=0.2 if alpha is None else float(alpha), prelu=slopes,
                                 want_f32=not want_split, want_split=want_split)


def double_conv_supported(dc, x):
    """unet_encoders.DoubleConv (BatchNorm2d -> conv -> PReLU -> conv -> PReLU -> PReLU) on the device path."""
    seq = dc.double_conv
    return (_device_path(x) and isinstance(seq[0], torch.nn.BatchNorm2d) and conv_supported(seq[1], *x.shape[-2:])
            and conv_supported(seq[3], *x.shape[-2:]) and x.shape[0] * x.shape[2] * x.shape[3] > 1)


def double_conv_forward(dc, x):
    """DoubleConv.forward (unet_encoders.py:52-66).  The BatchNorm -- batch statistics in train mode, as eval_seq.py runs these
    decoders -- is an affine map per channel, folded into the staging of the first convolution; the two trailing PReLUs are one
    leaky ReLU per channel (slope a1 * a2 where a1 > 0, else a1)."""
    bn, conv1, p1, conv2, p2, p3 = dc.double_conv
    x = x.contiguous()
    b = x.shape[0]
    if bn.training or not bn.track_running_stats:
        var, mean = torch.var_mean(x, dim=(0, 2, 3), unbiased=False)
        if bn.track_running_stats and bn.running_mean is not None:  # the side effect of a train-mode call (torch.nn.BatchNorm2d)
            n = x.numel() / x.shape[1]
            bn.num_batches_tracked += 1
            # momentum=None is the cumulative moving average: factor 1 / num_batches_tracked (torch/nn/modules/batchnorm.py)
            factor = bn.momentum if bn.momentum is not None else 1.0 / bn.num_batches_tracked.to(mean.dtype)
            bn.running_mean.lerp_(mean, factor)
            bn.running_var.lerp_(var * (n / max(n - 1, 1)), factor)
    else:
        var, mean = bn.running_var, bn.running_mean
    a = bn.weight.detach().float() * torch.rsqrt(var.float() + bn.eps)
    c = bn.bias.detach().float() - mean.float() * a
    xs = hipops.act_split(x, a.unsqueeze(0).expand(b, -1).contiguous(), shift=c.unsqueeze(0).expand(b, -1).contiguous())
    us = conv3x3(xs, conv1, slopes=p1.weight.detach().float().contiguous(), want_split=True)
    a2, a3 = p2.weight.detach().float(), p3.weight.detach().float()
    return conv3x3(us, conv2, slopes=torch.where(a2 > 0, a2 * a3, a2).contiguous())


def conv_lrelu_conv_supported(seq, x):
    """nn.Sequential(Conv2d 3x3, LeakyReLU, Conv2d 3x3): the CS-SFT heads of TriPlaneSFTfeat_Encoder (unet_encoders.py:262-270)."""
    return (_device_path(x) and len(seq) == 3 and isinstance(seq[1], torch.nn.LeakyReLU) and conv_supported(seq[0], *x.shape[-2:])
            and conv_supported(seq[2], *x.shape[-2:]))


def conv_lrelu_conv_forward(seq, xs):
    """`xs`: the SplitAct of the head's input (shared by the scale and the shift head)."""
    return conv3x3(conv3x3(xs, seq[0], alpha=seq[1].negative_slope, want_split=True), seq[2])
