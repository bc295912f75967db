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
  * a stride-2 conv2 (4 of the 48 convolutions of a trunk) runs on ``ia_conv2d_down_sx``: the same tiles with the point grid over every
    second pixel of the input window (r05; r03 / r04 evaluated it at stride 1 and sub-sampled -- still the route of the shapes the
    planner refuses).

Units in TRAIN mode (batch statistics: the e4e trunk under eval_seq.py's module modes) take the same route: the first BatchNorm's
affine map comes from the batch statistics of the unit's input (with the running-statistics update torch.nn.BatchNorm2d makes), the
second one is applied to conv2's fp32 result by the library's batch-norm kernel.  Layers from 8^2 up are covered (the planner cuts
the 8^2 / 16^2 layers between stream-K workers, conv_mfma.hip make_plan); CPU tensors and autograd take the unit's own ``torch.nn``
forward.

``conv_forward`` is the route of every other ``Conv2d`` of the encoders (layers.Conv2d): 1x1 layers and 3x3 layers with a handful of
input channels on the fp32 MFMA kernel (``ia_conv2d_mfma``; a stride-s 1x1 layer reads the sub-sampled input), 3x3 stride-1 layers
on ``ia_conv2d_mfma_sx``, 3x3 stride-2 layers on ``ia_conv2d_down_sx``.
"""
import torch

from ... import _runtime, hipops


class _UnitPack(_runtime.DeviceCache):
    """Kernel-side parameters of one residual unit, rebuilt when a parameter / buffer changes: packed weights and PReLU slopes under
    one key, the eval-mode affine maps of the two BatchNorms under another (train-mode calls move the running statistics)."""

    def __init__(self):
        self.key = self.bn_key = None

    def get(self, unit):
        conv1, prelu, conv2 = unit.res_layer[1], unit.res_layer[2], unit.res_layer[3]
        key = tuple((t.data_ptr(), t._version) for t in (conv1.weight, prelu.weight, conv2.weight)) + (conv1.weight.device,)
        if key != self.key:
            self.w1 = hipops.pack_conv_weight_split(conv1.weight.detach().float())
            self.w2 = hipops.pack_conv_weight_split(conv2.weight.detach().float())
            self.slopes = prelu.weight.detach().float().contiguous()
            self.key = key
        return self

    def eval_affines(self, unit, b):
        """(a1 rows, c1 rows, a2 rows, c2): the eval-mode affine maps of the unit's BatchNorms, the per-(batch, channel) arguments already
        expanded to `b` rows (a trunk pass is a chain of dependent launches: three tiny copies per unit and call were ~8 % of its nodes)."""
        bn1, bn2 = unit.res_layer[0], unit.res_layer[4]
        tensors = (bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var)
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if key != self.bn_key:
            self.affine = _affine(bn1, bn1.running_mean, bn1.running_var) + _affine(bn2, bn2.running_mean, bn2.running_var)
            self.rows = {}          # per batch size, all kept: a captured graph may hold the rows of one size while another size is in use
            self.bn_key = key
        if b not in self.rows:
            a1, c1, a2, c2 = self.affine
            self.rows[b] = (_rows(a1, b), _rows(c1, b), _rows(a2, b), c2)
        return self.rows[b]


def _affine(bn, mean, var):
    a = bn.weight.detach().float() * torch.rsqrt(var.detach().float() + bn.eps)
    return a.contiguous(), (bn.bias.detach().float() - mean.detach().float() * a).contiguous()


def batch_norm_affine(bn, x):
    """(scale, shift) per channel of the map ``bn`` applies to ``x``.  Train mode (or no running statistics): batch statistics, and
    the running-statistics update a train-mode call of torch.nn.BatchNorm2d makes; eval mode: the running statistics."""
    if bn.training or not bn.track_running_stats:
        var, mean = torch.var_mean(x, dim=(0, 2, 3), unbiased=False)
        if bn.track_running_stats and bn.running_mean is not None:
            n = x.numel() / x.shape[1]
            bn.num_batches_tracked += 1
            # momentum=None is the cumulative moving average: factor 1 / num_batches_tracked (torch/nn/modules/batchnorm.py)
            factor = bn.momentum if bn.momentum is not None else 1.0 / bn.num_batches_tracked.to(mean.dtype)
            bn.running_mean.lerp_(mean, factor)
            bn.running_var.lerp_(var * (n / max(n - 1, 1)), factor)
        return _affine(bn, mean, var)
    return _affine(bn, bn.running_mean, bn.running_var)


def _rows(v, b):
    return v.unsqueeze(0).expand(b, -1).contiguous()


BN_SPLIT_KERNEL = True      # train-mode BatchNorm -> split staging in two launches (ia_bn_train_split) instead of ~12 small ATen launches


def batch_norm_split(bn, x):
    """SplitAct of ``bn(x)`` for the convolution that follows (x contiguous fp32 on the device): batch statistics through
    ia_bn_train_split, running statistics through the cached affine map + ia_act_split."""
    b = x.shape[0]
    if BN_SPLIT_KERNEL and (bn.training or not bn.track_running_stats) and bn.momentum is not None:
        track = bn.track_running_stats and bn.running_mean is not None
        return hipops.bn_train_split(x, None if bn.weight is None else bn.weight.detach().float(), None if bn.bias is None else bn.bias.detach().float(),
                                     bn.running_mean if track else None, bn.running_var if track else None, bn.num_batches_tracked if track else None,
                                     bn.eps, bn.momentum)
    a, c = batch_norm_affine(bn, x)
    return hipops.act_split(x, _rows(a, b), shift=_rows(c, b))


def sx_size_ok(i, o, h, w):
    """3x3 stride-1 shapes ia_conv2d_mfma_sx takes: the 8-wave tile from 32^2 points and 64 outputs up, and whatever the library's own
    rule adds (ia_conv2d_sx_supported: 128 outputs and 8^2 points up -- the stream-K plans of the low-resolution layers)."""
    if i % 8 or o % 8 or w > 320:
        return False
    return (o >= 64 and h * w >= 1024) or hipops.conv_sx_supported(i, o, h, w, 3, False)


TRAIN_UNITS = True    # residual units in train mode (batch statistics) on the HIP convolutions too (False: the unit's torch.nn forward)
DOWN_TILES = True     # stride-2 3x3 layers on ia_conv2d_down_sx (False: at stride 1 and sub-sampled, the r03 / r04 route)


def unit_supported(unit, x):
    """True when the unit's two 3x3 convolutions can run on ia_conv2d_mfma_sx for this input."""
    if not (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()):
        return False
    return unit_shape_supported(unit, x.shape[0], x.shape[2], x.shape[3])


def unit_shape_supported(unit, b, h, w):
    """unit_supported for an fp32 device input of `b` images of h x w pixels (no tensor needed)."""
    if unit.training and (not TRAIN_UNITS or b * h * w <= 1):       # (torch.nn.BatchNorm2d raises on one value per channel)
        return False
    conv1, conv2 = unit.res_layer[1], unit.res_layer[3]
    i, o = conv1.in_channels, conv1.out_channels
    plain = all(c.padding == (1, 1) and c.dilation == (1, 1) and c.groups == 1 and c.padding_mode == 'zeros' for c in (conv1, conv2))
    return (plain and conv1.kernel_size == (3, 3) and conv2.kernel_size == (3, 3) and conv1.stride == (1, 1) and conv2.stride in ((1, 1), (2, 2))
            and conv1.bias is None and conv2.bias is None and sx_size_ok(i, o, h, w) and sx_size_ok(o, o, h, w)
            and conv2.in_channels == o and conv2.out_channels == o)


def _pack_of(unit):
    pack = _runtime.state(unit)
    if not hasattr(pack, 'trunk'):
        pack.trunk = _UnitPack()
    return pack.trunk.get(unit)


def next_unit_affine(unit, nxt, x):
    """The rows (scale, shift) [B, C] of `nxt`'s first BatchNorm when `nxt` will run unit_forward in EVAL mode on this unit's output
    (then the squeeze-and-excitation tail of `unit` writes nxt's staged input itself: SE_WRITES_NEXT_SPLIT), else None."""
    if not (SE_WRITES_NEXT_SPLIT and nxt is not None and len(nxt.res_layer) > 5 and len(unit.res_layer) > 5):
        return None
    bn1, bn2 = nxt.res_layer[0], nxt.res_layer[4]
    if bn1.training or bn2.training or not (bn1.track_running_stats and bn2.track_running_stats):
        return None
    s_ = unit.res_layer[3].stride[0]
    b, o = x.shape[0], unit.res_layer[3].out_channels
    oh, ow = (x.shape[2] - 1) // s_ + 1, (x.shape[3] - 1) // s_ + 1
    if o % 8 or nxt.res_layer[1].in_channels != o or not unit_shape_supported(nxt, b, oh, ow):
        return None
    return _pack_of(nxt).eval_affines(nxt, b)[:2]


SE_WRITES_NEXT_SPLIT = True     # eval-mode trunks: a unit's SE tail also writes the next unit's normalised split input (one launch fewer per unit)


def unit_forward(unit, x, xs=None, nxt=None):
    """bottleneck_IR_SE.forward (helpers.py:121-124) with the residual branch's convolutions on the HIP kernels.  `xs`: the unit's
    staged input (BatchNorm applied, split format) when the previous unit's tail wrote it; `nxt`: the unit that runs next (run_trunk).
    Returns the unit's output, or (output, staged input of nxt) when `nxt` is given."""
    p = _pack_of(unit)
    bn1, bn2 = unit.res_layer[0], unit.res_layer[4]
    b = x.shape[0]
    x = x.contiguous()
    stride = unit.res_layer[3].stride[0]
    batch_stats = bn1.training or bn2.training or not (bn1.track_running_stats and bn2.track_running_stats)
    if batch_stats:
        xs = batch_norm_split(bn1, x)
    elif xs is None:
        a1, c1 = p.eval_affines(unit, b)[:2]
        xs = hipops.act_split(x, a1, shift=c1)
    us = hipops.conv2d_mfma_sx(xs, p.w1, act='lrelu', prelu=p.slopes, want_f32=False, want_split=True)
    conv2, sub = hipops.conv2d_mfma_sx, stride == 2
    if stride == 2 and DOWN_TILES and hipops.conv_down_supported(b, us.shape[1], us.shape[1], *us.shape[2:]):
        conv2, sub = hipops.conv2d_down_sx, False
    if batch_stats:
        v = conv2(us, p.w2, act='linear')
        v = bn2(v[:, :, ::2, ::2].contiguous() if sub else v)
    else:
        a2, c2 = p.eval_affines(unit, b)[2:]
        v = conv2(us, p.w2, demod=a2, bias=c2, act='linear')
        if sub:
            v = v[:, :, ::2, ::2]
    if nxt is None:
        return se_tail(unit, v, x)
    return se_tail(unit, v, x, next_affine=next_unit_affine(unit, nxt, x), both=True)


def se_tail(unit, v, x, next_affine=None, both=False):
    """SEModule gate + shortcut + add of a residual unit (helpers.py:84-100, :121-124) in two launches (ia_se_gate); `v`: the residual
    branch in front of the gate (may be a strided view).  Units without a gate, CPU tensors and autograd take the ATen ops.
    `next_affine` = (scale rows, shift rows) [B, C]: also write split(out * scale + shift), the staged input of the next unit
    (ia_se_gate_split); `both`: return (out, that SplitAct or None)."""
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
        w1, w2 = se.fc1.weight.detach().float().reshape(-1, c).contiguous(), se.fc2.weight.detach().float().reshape(c, -1).contiguous()
        if next_affine is not None and c % 8 == 0:
            out, ys = hipops.se_gate_split(v, sc, w1, w2, next_affine[0], next_affine[1])
            return (out, ys) if both else out
        out = hipops.se_gate(v, sc, w1, w2)
        return (out, None) if both else out
    res = se(v) if se is not None else v
    out = res + short(x)
    return (out, None) if both else out


# ------------------------------------------------------------------ plain 3x3 convolutions of the UNet decoders / heads
def conv_supported(conv, h, w):
    """A torch.nn.Conv2d(3x3, stride 1, padding 1) whose shape ia_conv2d_mfma_sx takes (sx_size_ok)."""
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros'
            and sx_size_ok(conv.in_channels, conv.out_channels, h, w))


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
    xs = batch_norm_split(bn, x)
    us = conv3x3(xs, conv1, slopes=p1.weight.detach().float().contiguous(), want_split=True)
    st = _runtime.state(dc)
    key = (p2.weight.data_ptr(), p2.weight._version, p3.weight.data_ptr(), p3.weight._version)
    if getattr(st, 'slope_key', None) != key:      # (cached: three small launches per call otherwise)
        a2, a3 = p2.weight.detach().float(), p3.weight.detach().float()
        st.slopes, st.slope_key = torch.where(a2 > 0, a2 * a3, a2).contiguous(), key
    return conv3x3(us, conv2, slopes=st.slopes)


def conv_lrelu_conv_supported(seq, x):
    """nn.Sequential(Conv2d 3x3, LeakyReLU, Conv2d 3x3): the CS-SFT heads of TriPlaneSFTfeat_Encoder (unet_encoders.py:262-270)."""
    return (_device_path(x) and len(seq) == 3 and isinstance(seq[1], torch.nn.LeakyReLU) and conv_supported(seq[0], *x.shape[-2:])
            and conv_supported(seq[2], *x.shape[-2:]))


def conv_lrelu_conv_forward(seq, xs):
    """`xs`: the SplitAct of the head's input (shared by the scale and the shift head)."""
    return conv3x3(conv3x3(xs, seq[0], alpha=seq[1].negative_slope, want_split=True), seq[2])


# ------------------------------------------------------------------ every other Conv2d of the encoders (layers.Conv2d)
FP32_KERNEL_MAX_IN = 32     # 3x3 layers with at most this many input channels run on the fp32 MFMA kernel (image / UV input layers)


def _conv_route(conv, x):
    """Which kernel takes this torch.nn.Conv2d on this input: 'gemm' (1x1, any stride, no padding), 'patch' (k x k with stride k, no
    padding: a 1x1 layer on the space-to-depth image), 'sx' (3x3 stride 1 / 2, padding 1, split-format tile), 'f32' (3x3 stride 1 / 2,
    padding 1, few input channels), 'tiny' (3x3 stride 2 on 2^2 / 4^2 / 8^2 images: ia_conv3x3_s2_tiny, the last layers of the style heads) or
    None (library: in the one-shot decoders, the 7x7 stride-2 patch embeddings of the mix-transformer stages)."""
    if not (_device_path(x) and x.dim() == 4 and conv.groups == 1 and conv.dilation == (1, 1) and conv.padding_mode == 'zeros'
            and conv.weight.dtype == torch.float32 and conv.stride[0] == conv.stride[1]):
        return None
    b, i, h, w = x.shape
    o, s = conv.out_channels, conv.stride[0]
    if conv.kernel_size == (1, 1) and conv.padding == (0, 0):
        oh, ow = (h - 1) // s + 1, (w - 1) // s + 1
        return 'gemm' if oh * ow >= 16 else None       # (a 1x1 image is a matrix product: the squeeze-and-excitation gates have their own kernel)
    if conv.kernel_size == (s, s) and conv.padding == (0, 0) and s > 1:
        return 'patch' if (h // s) * (w // s) >= 16 else None
    if conv.kernel_size == (3, 3) and conv.padding == (1, 1) and s in (1, 2) and (s == 1 or (h % 2 == 0 and w % 2 == 0)):
        if s == 2 and hipops.conv_tiny_supported(i, o, h, w):      # (2^2 / 4^2 / 8^2 images: below the outputs ia_conv2d_down_sx starts at)
            return 'tiny'
        if sx_size_ok(i, o, h, w):
            return 'sx'
        if i <= FP32_KERNEL_MAX_IN and h * w >= 1024:
            return 'f32'
    return None


def conv_covered(conv, x):
    return _conv_route(conv, x) is not None


def _packed_f32(conv, as_1x1=False):
    st = _runtime.state(conv)
    key = (conv.weight.data_ptr(), conv.weight._version, conv.weight.device)
    if getattr(st, 'wf_key', None) != key:
        w = conv.weight.detach().float()
        st.wf = hipops.pack_conv_weight(w.reshape(w.shape[0], -1, 1, 1) if as_1x1 else w)
        st.wf_key = key
    return st.wf


def conv_forward(conv, x):
    """torch.nn.Conv2d.forward on the HIP kernels (see _conv_route).  A stride-2 3x3 layer with padding 1 on an even image is the
    stride-1 result at the even positions (the route of the shapes ia_conv2d_down_sx does not take)."""
    route = _conv_route(conv, x)
    s = conv.stride[0]
    bias = None if conv.bias is None else conv.bias.detach().float()
    if route == 'gemm':
        xin = x if s == 1 else x[:, :, ::s, ::s]
        return hipops.conv2d_mfma(xin.contiguous(), _packed_f32(conv), bias=bias, ksize=1)
    if route == 'patch':        # channels of the space-to-depth image in the weight's own (i, ky, kx) order
        b, i, h, w = x.shape
        oh, ow = h // s, w // s
        xin = x[:, :, :oh * s, :ow * s].reshape(b, i, oh, s, ow, s).permute(0, 1, 3, 5, 2, 4).reshape(b, i * s * s, oh, ow)
        return hipops.conv2d_mfma(xin.contiguous(), _packed_f32(conv, as_1x1=True), bias=bias, ksize=1)
    if route == 'tiny':
        return hipops.conv3x3_s2_tiny(x.contiguous(), conv.weight.detach(), bias=bias)
    if route == 'sx':
        xs = hipops.act_split(x.contiguous())
        if s == 2 and DOWN_TILES and hipops.conv_down_supported(x.shape[0], conv.in_channels, conv.out_channels, *x.shape[-2:]):
            return hipops.conv2d_down_sx(xs, packed_weight(conv), bias=bias)
        y = hipops.conv2d_mfma_sx(xs, packed_weight(conv), bias=bias)
    elif route == 'f32':
        y = hipops.conv2d_mfma(x.contiguous(), _packed_f32(conv), bias=bias, ksize=3)
    else:
        raise RuntimeError('conv_forward: this layer is not covered (ask conv_covered first)')
    return y if s == 1 else y[:, :, ::2, ::2].contiguous()


# ------------------------------------------------------------------ e4e style heads (e4e.GradualStyleBlock)
def style_head_supported(block, x):
    """GradualStyleBlock on the device path: every stage `Conv2d(3x3, stride 2, padding 1) + LeakyReLU` on ia_conv2d_down_sx or
    ia_conv3x3_s2_tiny."""
    if not (_device_path(x) and x.dim() == 4 and x.shape[-1] == x.shape[-2]):
        return False
    b, _, h, w = x.shape
    mods = list(block.convs)
    if len(mods) % 2 or not mods:
        return False
    for conv, act in zip(mods[0::2], mods[1::2]):
        if not (isinstance(conv, torch.nn.Conv2d) and isinstance(act, torch.nn.LeakyReLU) and conv.kernel_size == (3, 3) and conv.stride == (2, 2)
                and conv.padding == (1, 1) and conv.groups == 1 and conv.dilation == (1, 1) and conv.padding_mode == 'zeros'):
            return False
        i, o = conv.in_channels, conv.out_channels
        if not (hipops.conv_tiny_supported(i, o, h, w) or (i % 8 == 0 and o % 8 == 0 and h % 2 == 0 and hipops.conv_down_supported(b, i, o, h, w))):
            return False
        h, w = h // 2, w // 2
    return True


def style_head_forward(block, x, xs=None):
    """The convolution stack of a GradualStyleBlock (e4e.py:22-45) with the LeakyReLU in each convolution's epilogue and the split
    format handed from layer to layer (`xs`: the SplitAct of `x` when the caller shares it between the heads that read the same
    feature map): 9 launches for a 64^2 head instead of 18 + a split."""
    mods = list(block.convs)
    b, _, h, w = x.shape
    f32, split = x, xs
    pairs = list(zip(mods[0::2], mods[1::2]))
    for k, (conv, act) in enumerate(pairs):
        i, o = conv.in_channels, conv.out_channels
        bias = None if conv.bias is None else conv.bias.detach().float()
        alpha = float(act.negative_slope)
        if hipops.conv_tiny_supported(i, o, h, w):
            f32, split = hipops.conv3x3_s2_tiny(f32.contiguous(), conv.weight.detach(), bias=bias, act='lrelu', alpha=alpha), None
        else:
            if split is None:
                split = hipops.act_split(f32.contiguous())
            nh = h // 2
            nxt = pairs[k + 1][0] if k + 1 < len(pairs) else None
            keep_split = nxt is not None and not hipops.conv_tiny_supported(nxt.in_channels, nxt.out_channels, nh, nh)
            out = hipops.conv2d_down_sx(split, packed_weight(conv), bias=bias, act='lrelu', alpha=alpha, want_f32=not keep_split, want_split=keep_split)
            f32, split = (None, out) if keep_split else (out, None)
        h, w = h // 2, w // 2
    return f32
