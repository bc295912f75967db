"""UNet decoders with transformer-refined up-stages for the improved one-shot inversion (reference:
encoder_inversion/models/unet_transformer.py: ``TriPlanefeat_SegformerDecoder`` :255-337, ``TriPlaneSFTfeat_SegformerDecoder``
:340-445, ``UpLayer`` :527-547), the encoders `eval_updated_os.py` instantiates through uvnet_new.py:13-17.

Same IR-SE50 trunk, skip taps, heads and parameter names as the ConvGRU UNets of unet_encoders.py; what changes is the up-stage:
PixelShuffle -> concat skip -> ``transformer_block`` (4 / 4 / 3 / 3-or-2 ViT blocks at 1024 dims on the half-resolution token
grid) -> DoubleConv (-> ConvGRU when recurrent)."""
import torch
from torch import nn

from .mmseg.mix_transformer import transformer_block
from .unet_encoders import ConvGRU, DoubleConv, TriPlaneSFTfeat_Encoder, TriPlanefeat_Encoder


class UpLayer(nn.Module):
    def __init__(self, in_channels, out_channels, upscale_factor=2, use_gru=False, num_vit=0):
        super().__init__()
        self.up = nn.PixelShuffle(upscale_factor=upscale_factor)
        self.conv = DoubleConv(in_channels, out_channels)
        self.conv_gru = ConvGRU(out_channels, out_act_prelu=False) if use_gru else None
        self.use_vit = num_vit > 0
        self.transformer = transformer_block(in_chans=in_channels, num_vit=num_vit) if self.use_vit else None

    def forward(self, x1, x2=None, T=0, r=None):
        x = self.up(x1)
        if x2 is not None:
            x = torch.cat([x2, x], dim=1)
        if self.use_vit:
            x = self.transformer(x)
        x = self.conv(x)
        if self.conv_gru is None:
            return x
        return self.conv_gru(x.unflatten(0, (-1, T)), r, seq2seq=False)        # [B*T,C,H,W] -> ([B,C,H,W], state)


def _segformer_stages(net, use_gru, last_vit):
    """Replace the four up-stages a _UNetBase built by transformer-refined ones (same attribute names: up1 .. up4)."""
    net.up1 = UpLayer(1024, 512, upscale_factor=1, use_gru=use_gru, num_vit=4)
    net.up2 = UpLayer(384, 384, use_gru=use_gru, num_vit=4)
    net.up3 = UpLayer(224, 256, use_gru=use_gru, num_vit=3)
    net.up4 = UpLayer(128, 96, use_gru=use_gru, num_vit=last_vit)


class TriPlanefeat_SegformerDecoder(TriPlanefeat_Encoder):
    """Offsets for the first four neural-texture features: [32@32^2, 512@32^2, 512@64^2, 256@128^2]."""

    def __init__(self, inp_ch, sft_half=True, res=None, use_gru=False):
        super().__init__(inp_ch, res=res, use_gru=use_gru)
        self.sft_half = sft_half
        _segformer_stages(self, use_gru, last_vit=3)


class TriPlaneSFTfeat_SegformerDecoder(TriPlaneSFTfeat_Encoder):
    """CS-SFT (scale, shift) pairs for the static backbone at 16 .. 256^2."""

    def __init__(self, inp_ch, sft_half=True, res=None, use_gru=False):
        super().__init__(inp_ch, sft_half=sft_half, res=res, use_gru=use_gru)
        _segformer_stages(self, use_gru, last_vit=2)
