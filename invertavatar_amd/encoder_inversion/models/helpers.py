"""IR-SE50 building blocks (reference: encoder_inversion/models/helpers.py:17-124; ArcFace-style ResNet).
Module and parameter names follow the reference so that its checkpoints load by name."""
from collections import namedtuple

import torch
from torch.nn import AdaptiveAvgPool2d, BatchNorm2d, MaxPool2d, Module, PReLU, ReLU, Sequential, Sigmoid

from .layers import Conv2d


class Flatten(Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


def l2_norm(x, axis=1):
    return torch.div(x, torch.norm(x, 2, axis, True))


class Bottleneck(namedtuple('Block', ['in_channel', 'depth', 'stride'])):
    """(in_channel, depth, stride) of one residual unit."""


def get_block(in_channel, depth, num_units, stride=2):
    return [Bottleneck(in_channel, depth, stride)] + [Bottleneck(depth, depth, 1) for _ in range(num_units - 1)]


_STAGE_UNITS = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}


def get_blocks(num_layers):
    if num_layers not in _STAGE_UNITS:
        raise ValueError('Invalid number of layers: {}. Must be one of [50, 100, 152]'.format(num_layers))
    widths = ((64, 64), (64, 128), (128, 256), (256, 512))
    return [get_block(i, d, n) for (i, d), n in zip(widths, _STAGE_UNITS[num_layers])]


class SEModule(Module):
    """Squeeze-and-excitation gate: global pool -> 1x1 -> ReLU -> 1x1 -> sigmoid."""

    def __init__(self, channels, reduction):
        super().__init__()
        self.avg_pool = AdaptiveAvgPool2d(1)
        self.fc1 = Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        self.relu = ReLU(inplace=True)
        self.fc2 = Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)
        self.sigmoid = Sigmoid()

    def forward(self, x):
        return x * self.sigmoid(self.fc2(self.relu(self.fc1(self.avg_pool(x)))))


def _shortcut(in_channel, depth, stride):
    if in_channel == depth:
        return MaxPool2d(1, stride)
    return Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False), BatchNorm2d(depth))


class bottleneck_IR(Module):
    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self.shortcut_layer = _shortcut(in_channel, depth, stride)
        self.res_layer = Sequential(BatchNorm2d(in_channel), Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), PReLU(depth),
                                    Conv2d(depth, depth, (3, 3), stride, 1, bias=False), BatchNorm2d(depth))

    def forward(self, x):
        return self.res_layer(x) + self.shortcut_layer(x)


class bottleneck_IR_SE(Module):
    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self.shortcut_layer = _shortcut(in_channel, depth, stride)
        self.res_layer = Sequential(BatchNorm2d(in_channel), Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), PReLU(depth),
                                    Conv2d(depth, depth, (3, 3), stride, 1, bias=False), BatchNorm2d(depth), SEModule(depth, 16))

    def forward(self, x):
        return self.res_layer(x) + self.shortcut_layer(x)


def irse50_trunk(inp_ch):
    """(input_layer, body) of the IR-SE50 feature extractor shared by the three encoders."""
    input_layer = Sequential(Conv2d(inp_ch, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
    body = Sequential(*[bottleneck_IR_SE(u.in_channel, u.depth, u.stride) for stage in get_blocks(50) for u in stage])
    return input_layer, body


def face_pool_to(pool, x):
    """``pool(x)`` for an AdaptiveAvgPool2d whose target is exactly half the input (512^2 -> 256^2, the only case the encoders meet): the
    2x2 mean through avg_pool2d -- the same sums, and the ATen adaptive kernel takes 80 - 200 us for these few megabytes."""
    oh, ow = pool.output_size if isinstance(pool.output_size, tuple) else (pool.output_size, pool.output_size)
    if x.dim() == 4 and x.shape[-2] == 2 * oh and x.shape[-1] == 2 * ow:
        return torch.nn.functional.avg_pool2d(x, 2)
    return pool(x)


HIP_TRUNK = True      # eval-mode residual units on device tensors: their 3x3 convolutions through ia_conv2d_mfma_sx (trunk_hip.py)


def run_trunk(body, x, taps):
    """Run the residual units and collect the activations after the unit indices in `taps`."""
    from . import trunk_hip
    found = {}
    units = list(body._modules.values())
    xs = None          # the staged (BatchNorm + split) input of the next unit, when the previous unit's tail wrote it (eval-mode units)
    for i, unit in enumerate(units):
        if HIP_TRUNK and isinstance(unit, (bottleneck_IR, bottleneck_IR_SE)) and trunk_hip.unit_supported(unit, x):
            nxt = units[i + 1] if i + 1 < len(units) and isinstance(units[i + 1], bottleneck_IR_SE) else None
            x, xs = trunk_hip.unit_forward(unit, x, xs=xs, nxt=nxt) if nxt is not None else (trunk_hip.unit_forward(unit, x, xs=xs), None)
        elif HIP_TRUNK and isinstance(unit, bottleneck_IR_SE) and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled():
            x, xs = trunk_hip.se_tail(unit, unit.res_layer[:5](x), x), None      # library convolutions / BatchNorm, fused gate + shortcut + add
        else:
            x, xs = unit(x), None
        if i in taps:
            found[i] = x
    return x, [found[i] for i in taps]
