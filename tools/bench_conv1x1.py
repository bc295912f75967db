"""A/B of the ToRGB layer through the tiled ia_conv2d_mfma (ksize 1, stream-K + fix-up), the streaming ia_conv1x1 (+ ia_upfirdn2d for the skip
image) and the one-launch ia_torgb on the frame's shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from invertavatar_amd import hipops

SHAPES = [(512, 32, 64), (512, 96, 64), (256, 32, 128), (256, 96, 128), (128, 32, 256), (128, 96, 256), (256, 3, 256), (128, 3, 512),
          (512, 32, 32), (512, 96, 32), (512, 32, 16), (512, 96, 16), (512, 32, 8), (512, 96, 8), (512, 32, 4), (512, 96, 4)]


def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for i, o, r in SHAPES:
    x = torch.randn(1, i, r, r, device='cuda')
    s = torch.rand(1, i, device='cuda') + 0.5
    wk = hipops.pack_conv_weight(torch.randn(o, i, 1, 1, device='cuda') / i ** 0.5)
    bias, res = torch.randn(o, device='cuda'), torch.randn(1, o, r, r, device='cuda')
    t_old = bench(lambda: hipops.conv2d_mfma(x, wk, s, None, bias=bias, residual=res, ksize=1, act='linear', clamp=256))
    t_new = bench(lambda: hipops.conv1x1(x, wk, s, bias=bias, residual=res, clamp=256))
    nbytes = 4.0 * (x.numel() + 2 * res.numel())
    from invertavatar_amd.torch_utils.ops import upfirdn2d
    f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
    skip = torch.randn(1, o, r // 2, r // 2, device='cuda')
    t_two = bench(lambda: hipops.conv1x1(x, wk, s, bias=bias, residual=upfirdn2d.upsample2d(skip, f), clamp=256))
    t_one = bench(lambda: hipops.torgb(x, wk, s, bias=bias, skip=skip, skip_filter=f, clamp=256)) if hipops.torgb_supported(i, o, r, r, True) else float('nan')
    print(f'I={i:4d} O={o:3d} res={r:4d}  tiled {t_old:7.1f} us   streaming {t_new:7.1f} us ({nbytes / t_new / 1e3:7.1f} GB/s, '
          f'{2.0 * r * r * i * o / t_new / 1e6:6.1f} TF)   streaming + upsample2d {t_two:7.1f} us   ia_torgb (skip fused) {t_one:7.1f} us '
          f'({4.0 * (x.numel() + res.numel() + skip.numel()) / t_one / 1e3:7.1f} GB/s)', flush=True)
