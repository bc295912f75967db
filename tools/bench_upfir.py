"""The composed up-sampling layer of the SR head (ia_upconv2d_fir_sx: 32 -> 256 @128^2 -> 256^2, split-format output for the next layer)
timed alone with the epilogue the frame gives it (noise, bias, lrelu, clamp, next layer's styles).  IA_HIP_LIB selects a variant build."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import hipops

SHAPES = [(32, 256, 128), (32, 128, 256), (64, 256, 128)]


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


for i, o, r in SHAPES:
    if not hipops.upconv_fir_supported(1, i, o, r, r):
        print(f'I={i} O={o} res={r}: not covered')
        continue
    torch.manual_seed(0)
    x = torch.randn(1, i, r, r, device='cuda')
    s = torch.rand(1, i, device='cuda') + 0.5
    w = torch.randn(o, i, 3, 3, device='cuda')
    f = torch.tensor([1., 3., 3., 1.]); f = torch.outer(f, f); f = f / f.sum()
    wk = hipops.pack_conv_weight_split(hipops.compose_upfir_weight(w, f))
    xs = hipops.act_split(x, s)
    demod = torch.rand(1, o, device='cuda') + 0.5
    nz = torch.randn(4 * r * r, device='cuda')
    ns = torch.full((1,), 0.1, device='cuda')
    bias = torch.randn(o, device='cuda')
    sn = torch.rand(1, o, device='cuda') + 0.5
    fn = lambda: hipops.upconv_fir_sx(xs, wk, demod, nz, ns, bias, styles_next=sn, act='lrelu', gain=2 ** 0.5, clamp=256.0, split_for=object())
    t = bench(fn)
    out_bytes = 2 * 2 * o * 4 * r * r
    print(f'I={i:4d} O={o:4d} res={r:4d} | {t:7.1f} us   output {out_bytes / 1e6:.1f} MB -> {out_bytes / t / 1e6:.2f} TB/s of stores   '
          f'{3 * 4 * 2.0 * r * r * 9 * i * o / t / 1e6:6.0f} TF executed', flush=True)
