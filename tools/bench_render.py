"""Micro-benchmark of ia_render_rays at the BASELINE shape (B=1, 128^2 rays, 256^2 planes).  Prints a digest of every output
(image features, depth, weight sum, fine depths, index / order buffers) so that two builds of the kernel (IA_HIP_LIB=...) can be
compared bit for bit."""
import hashlib
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from invertavatar_amd import hipops, synthetic
from oracle import renderer as OR
torch.manual_seed(0)
B, nrr = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 128
frames = list(range(B))
planes = hipops.planes_channels_last(torch.randn(B, 3, 32, 256, 256, device='cuda') * 0.5)
cams = synthetic.camera_labels(frames)
ro, rd = OR.ray_sampler_zxc(cams[:, :16].view(-1, 4, 4), cams[:, 16:25].view(-1, 3, 3), nrr)
ro, rd = ro.cuda().contiguous(), rd.cuda().contiguous()
jit = synthetic.jitter(frames, nrr * nrr).squeeze(-1).cuda().contiguous()
dist = torch.norm(ro, dim=-1).mean().reshape(1)
w0, b0, w1, b1 = torch.randn(64, 32, device='cuda'), torch.randn(64, device='cuda') * .1, torch.randn(33, 64, device='cuda'), torch.randn(33, device='cuda') * .1
fn = lambda: hipops.render_rays(planes, ro, rd, jit, dist, w0, b0, w1, b1)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
fl = B * nrr * nrr * 96 * 2.0 * (32 * 64 + 64 * 33)       # 13.09 GFLOP per frame
print(f'B={B}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s algorithmic ({fl/ms/1e9/157.3*100:.1f}% of fp32 peak)')
rgb, depth, wsum, aux = hipops.render_rays(planes, ro, rd, jit, dist, w0, b0, w1, b1, debug=True)
h = hashlib.sha256()
for t in (rgb, depth, wsum, aux['z_fine'], aux['inds'], aux['order'], aux['w_coarse'], aux['sigma_coarse']):
    h.update(t.cpu().numpy().tobytes())
print('output digest', h.hexdigest()[:16], ' rgb sum %.9g' % rgb.double().sum().item())
print('  per output:', ' '.join(n + '=' + hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:8] for n, t in
                              (('rgb', rgb), ('depth', depth), ('wsum', wsum), ('z_fine', aux['z_fine']), ('inds', aux['inds']),
                               ('order', aux['order']), ('w_coarse', aux['w_coarse']), ('sigma_coarse', aux['sigma_coarse']))))
