"""Host cost of one hipGraph replay of the frame (no sync inside the loop) vs the GPU frame time."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import synthetic
from invertavatar_amd.graphed import GraphedSynthesis
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator

gen = TriPlaneGenerator(**synthetic.generator_kwargs('full')).eval().requires_grad_(False)
synthetic.fill_parameters(gen)
gen = gen.cuda()
with torch.no_grad():
    ws = gen.mapping(synthetic.latent(0, 1).cuda(), synthetic.conditioning_camera().cuda(), truncation_psi=0.7, truncation_cutoff=14)
    cams, uvs = synthetic.camera_labels([0]).cuda(), synthetic.uv_conditions([0]).cuda()
    jit = synthetic.jitter([0], 128 * 128).squeeze(-1).cuda()
    g = GraphedSynthesis(gen, batch=1, neural_rendering_resolution=128)
    g(ws, cams, uvs, jit)
    torch.cuda.synchronize()
    for label, sync_each in (('back-to-back', False), ('sync after each', True)):
        host = []
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        for k in range(30):
            t0 = time.perf_counter()
            g.graph.replay()
            host.append(time.perf_counter() - t0)
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        total = time.perf_counter() - t_all
        host.sort()
        print(f'{label}: replay() host time median {host[15] * 1e3:.3f} ms, min {host[0] * 1e3:.3f}, max {host[-1] * 1e3:.3f}; '
              f'wall per frame {total / 30 * 1e3:.3f} ms', flush=True)
