"""Super-resolution heads that turn the 32-channel neural render into RGB (reference:
training_avatar_texture/superresolution.py).  All heads are two StyleGAN2 synthesis blocks driven by the last
w; the default for 512^2 output is ``SuperresolutionHybrid8XDC`` (256 -> 128 channels, :263-289).  On device
tensors the blocks run on the fused MFMA convolution path of ``training.networks_stylegan2``."""
import numpy as np
import torch

from .. import _runtime
from ..torch_utils import misc, persistence
from ..torch_utils.ops import upfirdn2d
from ..training.networks_stylegan2 import Conv2dLayer, SynthesisBlock, SynthesisLayer, ToRGBLayer, _StyleBatcher


def _last_w(ws, n=3):
    return ws[:, -1:, :].repeat(1, n, 1)


def _fit(x, rgb, size, antialias):
    """Resize features + rgb to the head's input resolution when they differ (:281-285)."""
    if x.shape[-1] != size:
        kw = dict(size=(size, size), mode='bilinear', align_corners=False, antialias=antialias)
        x = torch.nn.functional.interpolate(x, **kw)
        rgb = torch.nn.functional.interpolate(rgb, **kw)
    return x, rgb


SKIP_IMAGE_STREAM = True     # block0's skip image on a side stream (see _TwoBlockHead.forward)
SKIP_UPSAMPLE_ON_SIDE = True     # ... and its up-sampled copy for block1's fused ToRGB epilogue


def _single_stream():
    """The generator's one-stream mode (bench.py's per-launch timing): no side stream here either."""
    from . import triplane_v20
    return triplane_v20.SINGLE_STREAM


class _TwoBlockHead(torch.nn.Module):
    """block0 then block1, both fed the last w three times."""

    def _setup(self, sr_num_fp16_res, sr_antialias, input_resolution):
        self.input_resolution = input_resolution
        self.sr_antialias = sr_antialias
        return sr_num_fp16_res > 0

    def _prepare_styles(self, ws3):
        rt = _runtime.state(self)          # (runtime state lives outside the module: deepcopy / pickle see parameters only)
        if not hasattr(rt, 'style_batcher'):
            rt.style_batcher = _StyleBatcher()
        if isinstance(self.block0, SynthesisBlock) and isinstance(self.block1, SynthesisBlock):
            rt.style_batcher.prepare([self.block0, self.block1], [0, 0], ws3.to(torch.float32))

    def hoist_styles(self, ws, prepared=False):
        """The head's styles depend only on ws: a caller that knows ws long before the rendered features exist (triplane_v20:
        at the top of the frame, on a side stream) can have them computed then; forward() with the same ws object uses them.
        `prepared`: the caller's frame-level style batch has parked this head's styles already (`style_blocks`)."""
        w3 = _last_w(ws)
        if not prepared:
            self._prepare_styles(w3)
        _runtime.state(self).hoisted = (ws, w3)

    def style_blocks(self):
        """The head's blocks for a frame-level style batch: every layer reads the LAST w (`_last_w`), or None when the head is not
        made of SynthesisBlocks."""
        if isinstance(self.block0, SynthesisBlock) and isinstance(self.block1, SynthesisBlock):
            return [self.block0, self.block1]
        return None

    def forward(self, rgb, x, ws, **block_kwargs):
        rt = _runtime.state(self)
        hoisted, rt.hoisted = getattr(rt, 'hoisted', None), None
        if hoisted is not None and hoisted[0] is ws:
            ws = hoisted[1]
        else:
            ws = _last_w(ws)
            self._prepare_styles(ws)
        x, rgb = _fit(x, rgb, self.input_resolution, self.sr_antialias)
        chain = {}
        if isinstance(self.block0, SynthesisBlock):     # block0.conv1 writes block1.conv0's operand format (fp16 plane(s), see hipops.SplitAct)
            from ..training import networks_stylegan2 as sg2
            next_half = bool(getattr(self.block1, 'use_fp16', False)) and ws.is_cuda and not sg2.FP16_BLOCKS_COMPUTE_FP32
            chain = dict(_next_conv=getattr(self.block1, 'conv0', None), _next_half=next_half)
        side = None
        if SKIP_IMAGE_STREAM and not _single_stream() and x.is_cuda and isinstance(self.block0, SynthesisBlock) and isinstance(self.block1, SynthesisBlock) \
                and not torch.is_grad_enabled():
            # block0's ToRGB (+ the up-sampling of the incoming image) feeds only the final image: on a side stream it runs beside
            # block1.conv0 instead of between the two largest convolutions of the frame (nothing else occupies the GPU here)
            side = getattr(rt, 'img_stream', None)
            if side is None or side.device != x.device:
                side = rt.img_stream = torch.cuda.Stream(device=x.device)
            chain['_img_stream'] = side
        rgb_in = rgb                     # read on the side stream: must outlive the join as well
        x0, rgb = self.block0(x, rgb, ws, **chain, **block_kwargs)
        last = dict(_x_unused=True) if isinstance(self.block1, SynthesisBlock) else {}      # block1's x has no reader: ToRGB in conv1's epilogue
        if side is not None:
            last['_img_wait'] = side
            if SKIP_UPSAMPLE_ON_SIDE and rgb is not None and getattr(self.block1, 'architecture', None) == 'skip':
                # block1's conv1 adds upsample2d(this image) in its fused ToRGB epilogue: up-sample it on the side stream as well
                # (one launch off the chain between the head's two largest convolutions)
                with torch.cuda.stream(side):
                    last['_skip_upsampled'] = upfirdn2d.upsample2d(rgb, self.block1.resample_filter).float().contiguous()
                last['_skip_upsampled'].record_stream(torch.cuda.current_stream(x0.device if torch.is_tensor(x0) else ws.device))
        x, rgb = self.block1(x0, rgb, ws, **last, **block_kwargs)
        del x0, rgb_in                   # (block0's features and input image stayed alive until block1 had joined the side stream)
        return rgb


@persistence.persistent_class
class SuperresolutionHybrid8X(_TwoBlockHead):
    """128^2 -> 512^2 with 128 / 64 channels (:28-55)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 512
        fp16 = self._setup(sr_num_fp16_res, sr_antialias, 128)
        clamp = 256 if fp16 else None
        self.block0 = SynthesisBlock(channels, 128, w_dim=512, resolution=256, img_channels=3, is_last=False, use_fp16=fp16,
                                     conv_clamp=clamp, **block_kwargs)
        self.block1 = SynthesisBlock(128, 64, w_dim=512, resolution=512, img_channels=3, is_last=True, use_fp16=fp16,
                                     conv_clamp=clamp, **block_kwargs)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))


@persistence.persistent_class
class SuperresolutionHybrid8XDC(_TwoBlockHead):
    """128^2 -> 512^2 with 256 / 128 channels: the head the 512^2 avatars are trained with (:263-289)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 512
        fp16 = self._setup(sr_num_fp16_res, sr_antialias, 128)
        clamp = 256 if fp16 else None
        self.block0 = SynthesisBlock(channels, 256, w_dim=512, resolution=256, img_channels=3, is_last=False, use_fp16=fp16,
                                     conv_clamp=clamp, **block_kwargs)
        self.block1 = SynthesisBlock(256, 128, w_dim=512, resolution=512, img_channels=3, is_last=True, use_fp16=fp16,
                                     conv_clamp=clamp, **block_kwargs)


@persistence.persistent_class
class SynthesisBlockNoUp(torch.nn.Module):
    """Synthesis block whose first convolution does not up-sample (:157-256): used by the 4X/2X heads."""

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
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp,
                                        channels_last=self.channels_last, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp,
                                    channels_last=self.channels_last, **layer_kwargs)
        self.num_conv += 1
        if is_last or architecture == 'skip':
            self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp, channels_last=self.channels_last)
            self.num_torgb += 1
        if in_channels != 0 and architecture == 'resnet':
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2, resample_filter=resample_filter,
                                    channels_last=self.channels_last)

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, **layer_kwargs):
        misc.assert_shape(ws, [None, self.num_conv + self.num_torgb, self.w_dim])
        w_iter = iter(ws.unbind(dim=1))
        if fused_modconv is None:
            fused_modconv = self.fused_modconv_default
        if fused_modconv == 'inference_only':
            fused_modconv = (not self.training)
        if self.in_channels == 0:
            x = self.const.to(torch.float32).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        else:
            misc.assert_shape(x, [None, self.in_channels, self.resolution, self.resolution])
            x = x.to(torch.float32)
            if self.architecture == 'resnet':
                y = self.skip(x, gain=np.sqrt(0.5))
                x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
                x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, gain=np.sqrt(0.5), **layer_kwargs)
                x = y.add_(x)
            else:
                x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
                x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        if self.is_last or self.architecture == 'skip':
            img = self.torgb(x, next(w_iter), fused_modconv=fused_modconv, residual=img)
            img = img.to(dtype=torch.float32, memory_format=torch.contiguous_format)
        return x, img


@persistence.persistent_class
class SuperresolutionHybrid4X(_TwoBlockHead):
    """128^2 -> 256^2 (:61-87)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 256
        fp16 = self._setup(sr_num_fp16_res, sr_antialias, 128)
        clamp = 256 if fp16 else None
        self.block0 = SynthesisBlockNoUp(channels, 128, w_dim=512, resolution=128, img_channels=3, is_last=False, use_fp16=fp16,
                                         conv_clamp=clamp, **block_kwargs)
        self.block1 = SynthesisBlock(128, 64, w_dim=512, resolution=256, img_channels=3, is_last=True, use_fp16=fp16,
                                     conv_clamp=clamp, **block_kwargs)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))

    def forward(self, rgb, x, ws, **block_kwargs):
        ws = _last_w(ws)
        if x.shape[-1] < self.input_resolution:   # this head only ever up-samples its input (:79)
            x, rgb = _fit(x, rgb, self.input_resolution, self.sr_antialias)
        chain = {}
        if isinstance(self.block0, SynthesisBlock):     # block0.conv1 writes block1.conv0's operand format (fp16 plane(s), see hipops.SplitAct)
            from ..training import networks_stylegan2 as sg2
            next_half = bool(getattr(self.block1, 'use_fp16', False)) and ws.is_cuda and not sg2.FP16_BLOCKS_COMPUTE_FP32
            chain = dict(_next_conv=getattr(self.block1, 'conv0', None), _next_half=next_half)
        x, rgb = self.block0(x, rgb, ws, **chain, **block_kwargs)
        last = dict(_x_unused=True) if isinstance(self.block1, SynthesisBlock) else {}      # block1's x has no reader: ToRGB in conv1's epilogue
        x, rgb = self.block1(x, rgb, ws, **last, **block_kwargs)
        return rgb


@persistence.persistent_class
class SuperresolutionHybrid2X(_TwoBlockHead):
    """64^2 -> 128^2 (:93-120)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, sr_antialias, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, **block_kwargs):
        super().__init__()
        assert img_resolution == 128
        fp16 = self._setup(sr_num_fp16_res, sr_antialias, 64)
        clamp = 256 if fp16 else None
        self.block0 = SynthesisBlockNoUp(channels, 128, w_dim=512, resolution=64, img_channels=3, is_last=False, use_fp16=fp16,
                                         conv_clamp=clamp, **block_kwargs)
        self.block1 = SynthesisBlock(128, 64, w_dim=512, resolution=128, img_channels=3, is_last=True, use_fp16=fp16,
                                     conv_clamp=clamp, **block_kwargs)
        self.register_buffer('resample_filter', upfirdn2d.setup_filter([1, 3, 3, 1]))


@persistence.persistent_class
class SuperresolutionHybridDeepfp32(SuperresolutionHybrid4X):
    """Legacy 256^2 head: as 4X but the input resize is not anti-aliased and takes no sr_antialias (:126-152)."""

    def __init__(self, channels, img_resolution, sr_num_fp16_res, num_fp16_res=4, conv_clamp=None, channel_base=None,
                 channel_max=None, **block_kwargs):
        super().__init__(channels, img_resolution, sr_num_fp16_res, sr_antialias=False, **block_kwargs)
