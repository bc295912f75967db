"""Upper bound for running the frame's three backbones as ONE grouped launch per layer (network index as the batch index): one backbone
at batch 3 on one stream against the same backbone three times at batch 1 on three streams (what the frame does today), both captured."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import synthetic
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator

gen = TriPlaneGenerator(**synthetic.generator_kwargs('full')).eval().requires_grad_(False)
synthetic.fill_parameters(gen)
gen = gen.cuda()
net = gen.texture_backbone


def timed(graph, n=50):
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    ws1 = gen.mapping(synthetic.latent(0, 1).cuda(), synthetic.conditioning_camera().cuda(), truncation_psi=0.7, truncation_cutoff=14)
    ws3 = ws1.repeat(3, 1, 1).contiguous()
    run = lambda ws: net.synthesis(ws, cond_list=None, return_list=True, noise_mode='const')
    for _ in range(3):
        run(ws1); run(ws3)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]

    def three_streams():
        main = torch.cuda.current_stream()
        outs = []
        for s in streams:
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(run(ws1))
        for s in streams:
            main.wait_stream(s)
        return outs

    for name, fn in (('batch 1 on one stream', lambda: run(ws1)), ('3 x batch 1 on three streams', three_streams), ('batch 3 on one stream', lambda: run(ws3))):
        g = torch.cuda.CUDAGraph()
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            keep = fn()
        print(f'{name:32s} {timed(g):8.1f} us', flush=True)
