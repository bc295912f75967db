"""Probe: replay time of the static / texture backbone and the face-backbone head captured alone and forked in one graph."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch

from invertavatar_amd import synthetic
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator

gen = TriPlaneGenerator(**synthetic.generator_kwargs('full')).eval().requires_grad_(False)
synthetic.fill_parameters(gen)
gen = gen.cuda()
kw = dict(noise_mode='const')
with torch.no_grad():
    ws = gen.mapping(synthetic.latent(0, 1).cuda(), synthetic.conditioning_camera().cuda(), truncation_psi=0.7, truncation_cutoff=14)
    parts = {
        'static': lambda: gen.backbone.synthesis(ws, cond_list=None, return_list=True, **kw),
        'texture': lambda: gen.texture_backbone.synthesis(ws, cond_list=None, return_list=True, **kw),
        'face_head': lambda: gen.face_backbone.synthesis.forward_head(ws, **kw),
    }
    streams = {k: torch.cuda.Stream() for k in parts}
    cap = torch.cuda.Stream()

    def forked(names):
        main = torch.cuda.current_stream()
        outs = []
        for k in names:
            streams[k].wait_stream(main)
            with torch.cuda.stream(streams[k]):
                outs.append(parts[k]())
        for k in names:
            main.wait_stream(streams[k])
        return outs

    def measure(label, fn):
        with torch.cuda.stream(cap):
            keep = fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap):
                keep = fn()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                g.replay()
            torch.cuda.synchronize()
            print(f'{label:34s} {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us', flush=True)
        return keep

    for k in parts:
        measure(k + ' alone', parts[k])
    measure('static + texture forked', lambda: forked(['static', 'texture']))
    measure('static + texture + face_head forked', lambda: forked(['static', 'texture', 'face_head']))
    measure('the three in program order', lambda: [parts[k]() for k in parts])
