"""conv2d / conv_transpose2d entry points (reference: torch_utils/ops/conv2d_gradfix.py:37-45).

The reference's arbitrary-order-gradient custom op is a training feature and disabled by
default (`enabled = False`, :23); inference goes straight to the library convolution, which is
what these wrappers do.  The dense 3x3 layers of the generator do not come through here on the
GPU -- they use the MFMA kernels behind ``modulated_conv2d``."""
import contextlib

import torch

enabled = False
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    global weight_gradients_disabled
    old = weight_gradients_disabled
    if disable:
        weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                      dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)
