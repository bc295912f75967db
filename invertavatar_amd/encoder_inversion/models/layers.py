"""``Conv2d`` of the inversion encoders: ``torch.nn.Conv2d`` (same parameters, same state-dict keys) whose inference path on device
tensors runs the convolution kernels of libia_hip.so instead of the library (trunk_hip.conv_forward).  Autograd, CPU tensors and the
shapes the kernels do not take (grouped / dilated / 7x7 / non-square strides) run ``torch.nn.Conv2d.forward``.

Reference: the encoders build ``torch.nn.Conv2d`` directly (encoder_inversion/models/helpers.py:36-124, e4e.py:22-45,
unet_encoders.py:17-66 and :150-362)."""
import torch

HIP_CONVS = True      # False: every layer through torch.nn.Conv2d.forward (A/B switch: bench.py --set ...layers.HIP_CONVS=False)


class Conv2d(torch.nn.Conv2d):
    def forward(self, x):
        if HIP_CONVS and x.is_cuda:
            from . import trunk_hip
            if trunk_hip.conv_covered(self, x):
                return trunk_hip.conv_forward(self, x)
        return super().forward(x)
