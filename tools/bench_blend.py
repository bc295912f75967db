"""Times ia_blend_planes at the BASELINE shape (stitch 32 x 256^2, static planes 96 x 256^2, bbox 128^2)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from invertavatar_amd import hipops

stitch, alpha, sta = torch.randn(1, 32, 256, 256, device='cuda'), torch.rand(1, 256, 256, device='cuda'), torch.randn(1, 96, 256, 256, device='cuda')
fn = lambda: hipops.blend_planes(stitch, alpha, sta, [57, 185, 64, 192])   # noqa: E731
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
print(f'blend_planes: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us')
