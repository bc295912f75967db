"""The small 3x3 layers of a network head (4^2 .. 16^2 and 16^2 -> 32^2): ia_conv2d_small vs the stream-K kernel + fix-up."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import hipops


def bench(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for i, o, r, tr in [(512, 512, 4, 0), (512, 512, 4, 1), (512, 512, 8, 0), (512, 512, 8, 1), (512, 512, 16, 0), (512, 512, 16, 1)]:
    x = torch.randn(1, i, r, r, device='cuda')
    s = torch.rand(1, i, device='cuda') + 0.5
    wk = hipops.pack_conv_weight(torch.randn(o, i, 3, 3, device='cuda'))
    d = torch.rand(1, o, device='cuda')
    out = {}
    for flag in (True, False):
        hipops.SMALL_CONV = flag
        out[flag] = bench(lambda: hipops.conv2d_mfma(x, wk, s, d, ksize=3, transposed=bool(tr)))
    hipops.SMALL_CONV = True
    fl = 2.0 * r * r * 9 * i * o
    print(f'I={i} O={o} res={r:3d} tr={tr}: small {out[True]:6.1f} us ({fl / out[True] / 1e6:5.1f} TF)   stream-K + fix-up {out[False]:6.1f} us', flush=True)
