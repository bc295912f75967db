"""e4e W+ encoder and the two ConvGRU UNets of InvertAvatar (reference: encoder_inversion/models/{uvnet,e4e,unet_encoders,helpers}.py)."""
