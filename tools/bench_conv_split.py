"""A/B of the register-staged (ia_conv2d_mfma_s) and the LDS-DMA (ia_conv2d_mfma_sx) fp16-pair convolutions on the large layers."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from invertavatar_amd import hipops

SHAPES = [(256, 256, 256, 0), (128, 128, 512, 0), (512, 512, 64, 0), (256, 256, 128, 0), (128, 128, 256, 0)]


def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for i, o, r, tr in SHAPES:
    x = torch.randn(1, i, r, r, device='cuda')
    s = torch.rand(1, i, device='cuda') + 0.5
    wk = hipops.pack_conv_weight_split(torch.randn(o, i, 3, 3, device='cuda'))
    xs = hipops.act_split(x, s)
    t_old = bench(lambda: hipops.conv2d_mfma(x, wk, styles=s, ksize=3, transposed=bool(tr)))
    t_new = bench(lambda: hipops.conv2d_mfma_sx(xs, wk, transposed=bool(tr)))
    t_cvt = bench(lambda: hipops.act_split(x, s))
    fl = 2.0 * r * r * 9 * i * o
    print(f'I={i:4d} O={o:4d} res={r:4d} tr={tr}  staged {t_old:7.1f} us ({3*fl/t_old/1e6:6.0f} TF exec)   dma {t_new:7.1f} us ({3*fl/t_new/1e6:6.0f} TF exec, '
          f'{3*fl/t_new/1e6/2500:.3f} of peak)   act_split {t_cvt:6.1f} us', flush=True)
