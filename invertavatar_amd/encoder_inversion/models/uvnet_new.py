"""Improved one-shot inversion network of `eval_updated_os.py` (reference: encoder_inversion/models/uvnet_new.py:13-141).

The same wrapper as uvnet.inversionNet -- e4e W+ encoder, texture UNet on the UV-space residual, tri-plane UNet producing CS-SFT
conditions, frozen TriPlaneGenerator -- with the two UNets replaced by their transformer-refined, non-recurrent variants
(unet_transformer.py).  Only the one-shot ``forward`` exists in the reference file; it is inherited unchanged."""
from torch import nn

from . import unet_transformer, uvnet


class improved_os_unet_encoder(nn.Module):
    def __init__(self, encoding_texture=False, encoding_triplane=False):
        super().__init__()
        self.texture_unet = unet_transformer.TriPlanefeat_SegformerDecoder(inp_ch=7, res=256) if encoding_texture else None
        self.triplane_unet = unet_transformer.TriPlaneSFTfeat_SegformerDecoder(inp_ch=6, res=256) if encoding_triplane else None

    def forward(self, x):
        raise NotImplementedError


class inversionNet(uvnet.inversionNet):
    def __init__(self, G_kwargs=None, generator=None, encoding_texture=True, encoding_triplane=False):
        super().__init__(G_kwargs=G_kwargs, generator=generator, encoding_texture=False, encoding_triplane=False)
        self.unet_encoder = improved_os_unet_encoder(encoding_texture=encoding_texture, encoding_triplane=encoding_triplane)

    AR_eval_forward = None      # (the reference's uvnet_new.inversionNet has no incremental mode: its UNets are not recurrent)
