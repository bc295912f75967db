"""Per-launch table of one eager BASELINE frame: every fused stage with its shape, duration and algorithmic rate.
(Launches are timed with events on their own streams; concurrent streams overlap, so the column does not sum to the frame.)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch

from invertavatar_amd import hipops, synthetic
from invertavatar_amd.training_avatar_texture import triplane_v20
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator

triplane_v20.SINGLE_STREAM = '--single' in sys.argv      # program order on one stream: clean per-launch durations

gen = TriPlaneGenerator(**synthetic.generator_kwargs('full')).eval().requires_grad_(False)
synthetic.fill_parameters(gen)
gen = gen.cuda()
with torch.no_grad():
    ws = gen.mapping(synthetic.latent(0, 1).cuda(), synthetic.conditioning_camera().cuda(), truncation_psi=0.7, truncation_cutoff=14)
    cams, uvs = synthetic.camera_labels([0]).cuda(), synthetic.uv_conditions([0]).cuda()
    jit = synthetic.jitter([0], 128 * 128).squeeze(-1).cuda()
    run = lambda: gen.synthesis(ws, cams, {'uvcoords_image': uvs}, neural_rendering_resolution=128, noise_mode='const',
                                evaluation=True, jitter=jit)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    reps = 5
    hipops.PROFILE = []
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    recs, hipops.PROFILE = hipops.PROFILE, None
n = len(recs) // reps
tot = 0.0
for j in range(n):
    name, flops, nbytes, _, _, desc = recs[j]
    us = sum(recs[j + r * n][3].elapsed_time(recs[j + r * n][4]) for r in range(reps)) / reps * 1e3
    tot += us
    print(f'{j:3d} {name:20s} {desc:34s} {us:8.1f} us  {flops / us / 1e6:7.1f} TF  {nbytes / us / 1e3:8.1f} GB/s')
print(f'sum {tot:.1f} us')
