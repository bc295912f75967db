"""Frames in flight: the headline step through graphed.FramePipeline at depth 1, 2, 3 (frame k replayed on stream k % depth with its own
static buffers) against the single captured graph.  python tools/bench_pipeline.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

import bench
from invertavatar_amd import synthetic
from invertavatar_amd.graphed import FramePipeline, GraphedSynthesis
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
bench.DEV, bench.NRR = torch.device('cuda', 0), 128
gen = TriPlaneGenerator(**synthetic.generator_kwargs('full')).eval().requires_grad_(False)
synthetic.fill_parameters(gen)
gen = gen.to(bench.DEV)
with torch.no_grad():
    wl = bench.Workload(gen, 1, 0, 1, n_sets=16)
    single = GraphedSynthesis(gen, batch=1, neural_rendering_resolution=128)
    ref = wl.replay(single, 0)['image'].clone()

    def run(fn, finish=lambda: None):
        for k in range(10):
            fn(k)
        finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            fn(k)
        finish()
        torch.cuda.synchronize()
        return steps / (time.perf_counter() - t0)
    print(f'single graph: {run(lambda k: wl.replay(single, k % 16)):.1f} frames/s', flush=True)
    for depth in (1, 2, 3):
        pipe = FramePipeline(gen, depth=depth, batch=1, neural_rendering_resolution=128).capture(wl.ws, wl.cams[0], wl.uvs[0], wl.jits[0])
        out, ev, _ = pipe.submit(wl.ws, wl.cams[0], wl.uvs[0], wl.jits[0])
        ev.synchronize()
        err = (out['image'] - ref).abs().max().item()
        fps = run(lambda k: pipe.submit(wl.ws, wl.cams[k % 16], wl.uvs[k % 16], wl.jits[k % 16]), pipe.drain)
        print(f'pipeline depth {depth}: {fps:.1f} frames/s (frame 0 equals the single graph: max |d| = {err:.1e})', flush=True)
