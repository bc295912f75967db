"""Times ia_rasterize_level on the four levels of the BASELINE model (texture pyramid 32ch@32, 512@32, 512@64, 256@128)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import hipops, synthetic

uv_smooth = synthetic.uv_conditions([3]).cuda().contiguous()
uv_sil = uv_smooth.clone()                 # UV = 0 outside the face, as ia_uv_rasterize writes it: silhouette footprints see two texel clusters
uv_sil[..., 0][uv_sil[..., 2] < 0.5] = 0.0
uv_sil[..., 1][uv_sil[..., 2] < 0.5] = 0.0
for (name, uv), (c, r) in [(u, l) for u in (('smooth', uv_smooth), ('silhouette', uv_sil)) for l in [(32, 32), (512, 32), (512, 64), (256, 128)]]:
    upper = uv[..., 2].clamp(0, 1).contiguous()
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
    print(f'{name:10s} C={c:4d} res={r:4d}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us', flush=True)
