"""Captured e4e encode (stage A of the inversion) under the route switches of the encoder convolutions:
layers.HIP_CONVS (every plain Conv2d: style heads, laterals, input layer, shortcuts) and trunk_hip.TRAIN_UNITS (the train-mode residual
units of the e4e trunk).  GPU milliseconds of one graph replay, best of 5.  python tools/ab_encode.py"""
import itertools
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import eval_seq, synthetic
from invertavatar_amd.encoder_inversion.models import layers, trunk_hip
from invertavatar_amd.encoder_inversion.models.uvnet import inversionNet
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator

gen = TriPlaneGenerator(**synthetic.generator_kwargs('full')).eval().requires_grad_(False)
synthetic.fill_parameters(gen)
net = inversionNet(generator=gen, encoding_triplane=True, encoding_texture=True).requires_grad_(False)
synthetic.fill_encoder_parameters(net)
net = eval_seq.set_eval_seq_modes(net.cuda())
image = synthetic.source_frames(7, 4)[:1].cuda()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


with torch.no_grad():
    for convs, units, down in itertools.product((True, False), (True, False), (True, False)):
        if not down and not (convs or units):
            continue
        layers.HIP_CONVS, trunk_hip.TRAIN_UNITS, trunk_hip.DOWN_TILES = convs, units, down
        enc = eval_seq.GraphedEncode(net, image)
        t = timed(lambda: enc(image))
        e = timed(lambda: net.encode(image))
        print(f'HIP_CONVS {convs!s:5s} TRAIN_UNITS {units!s:5s} DOWN_TILES {down!s:5s}: captured {t:6.2f} ms   eager {e:6.2f} ms')
