"""Per-phase cycle split of ia_render_rays: a build with -DIA_RENDER_TRACE=1 (tools/_variants/libia_render_trace.so)
lets workgroup 0 stamp s_memtime at the phase boundaries of its first 8 rays per wave; this prints the mean cycles per phase.
  build (CPU):  hipcc ... -DIA_RENDER_TRACE=1 -c csrc/render_rays.hip ; link with the other objects      run (GPU):  python tools/trace_render.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('IA_HIP_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libia_render_trace.so'))
import torch

from invertavatar_amd import hipops, synthetic
from oracle import renderer as OR

torch.manual_seed(0)
nrr = 128
planes = hipops.planes_channels_last(torch.randn(1, 3, 32, 256, 256, device='cuda') * 0.5)
cams = synthetic.camera_labels([0])
ro, rd = OR.ray_sampler_zxc(cams[:, :16].view(-1, 4, 4), cams[:, 16:25].view(-1, 3, 3), nrr)
ro, rd = ro.cuda().contiguous(), rd.cuda().contiguous()
jit = synthetic.jitter([0], nrr * nrr).squeeze(-1).cuda().contiguous()
dist = torch.norm(ro, dim=-1).mean().reshape(1)
w0, b0, w1, b1 = torch.randn(64, 32, device='cuda'), torch.randn(64, device='cuda') * .1, torch.randn(33, 64, device='cuda'), torch.randn(33, device='cuda') * .1
for _ in range(3):
    hipops.render_rays(planes, ro, rd, jit, dist, w0, b0, w1, b1)
rgb, depth, wsum, aux = hipops.render_rays(planes, ro, rd, jit, dist, w0, b0, w1, b1, debug=True)
torch.cuda.synchronize()
t = aux['sigma_coarse'].reshape(-1)[:8 * 8 * 32 * 2].view(torch.int64).reshape(8, 8, 32).cpu().double()      # [wave][ray][slot]
t = t[:, 2:7]                                                   # steady-state rays of every wave
names = {}
for g in range(3):
    for start, tag in ((4 * g, 'coarse'), (15 + 4 * g, 'fine')):
        names[f'{tag} g{g}: gather + reduce + LDS hand-over'] = (start, start + 1)
        names[f'{tag} g{g}: prefetch issue + layer 1 (32 MFMA) + 16 softplus'] = (start + 1, start + 2)
        names[f'{tag} g{g}: density row'] = (start + 2, start + 3)
        names[f'{tag} g{g}: layer 2 colours (32 MFMA) + 8 sigmoid + store'] = (start + 3, start + 4)
names['coarse ray march (weights)'] = (12, 13)
names['importance resampling'] = (13, 14)
names['merge'] = (14, 15)
names['compositing (6 groups of 16)'] = (28, 29)
names['wave_sync after the fine pass'] = (27, 28)
names['row sums + stores'] = (29, 30)
total = (t[:, :, 30] - t[:, :, 0]).mean().item()
agg = {}
for k, (a, b) in names.items():
    d = (t[:, :, b] - t[:, :, a]).mean().item()
    key = k.split(': ')[1] if ': ' in k else k
    agg[key] = agg.get(key, 0.0) + d
print(f'cycles per ray (mean over 8 waves x 5 rays of workgroup 0): {total:.0f}')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f'  {v:9.0f}  {100 * v / total:5.1f} %  {k}')
print(f'  {total - sum(agg.values()):9.0f}         unaccounted (stamps, loop)')
