"""Times ia_rasterize_level on the four levels of the BASELINE model (texture pyramid 32ch@32, 512@32, 512@64, 256@128)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import hipops, synthetic

uv = synthetic.uv_conditions([3]).cuda().contiguous()
upper = uv[..., 2].clamp(0, 1).contiguous()
for c, r in [(32, 32), (512, 32), (512, 64), (256, 128)]:
    tex = torch.randn(1, c, r, r, device='cuda')
    sta = torch.randn(1, c, r, r, device='cuda')
    bbox = [round(v * r / 256) for v in (57, 185, 64, 192)]
    tcl = hipops.channels_last_copy(tex)
    fn = lambda: hipops.rasterize_level(tex, uv, upper, sta, bbox, r, tex_cl=tcl)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f'C={c:4d} res={r:4d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us', flush=True)
