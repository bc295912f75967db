"""ia_torgb on the frame's large-image shapes, with the skip image, with a residual and bare:  python tools/bench_torgb.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from invertavatar_amd import hipops
from invertavatar_amd.torch_utils.ops import upfirdn2d

SHAPES = [(128, 32, 256), (128, 96, 256), (256, 32, 128), (256, 96, 128), (512, 32, 64), (512, 96, 64)]


def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
for i, o, r in SHAPES:
    x = torch.randn(1, i, r, r, device='cuda')
    s = torch.rand(1, i, device='cuda') + 0.5
    wk = hipops.pack_conv_weight(torch.randn(o, i, 1, 1, device='cuda') / i ** 0.5)
    bias, res, skip = torch.randn(o, device='cuda'), torch.randn(1, o, r, r, device='cuda'), torch.randn(1, o, r // 2, r // 2, device='cuda')
    t_skip = bench(lambda: hipops.torgb(x, wk, s, bias=bias, skip=skip, skip_filter=f, clamp=256))
    t_res = bench(lambda: hipops.torgb(x, wk, s, bias=bias, residual=res, clamp=256))
    t_bare = bench(lambda: hipops.torgb(x, wk, s, bias=bias, clamp=256))
    t_copy = bench(lambda: x.sum())
    mb = 4e-6 * (x.numel() + res.numel())
    print(f'I={i:4d} O={o:3d} res={r:4d}  skip {t_skip:6.1f} us ({mb / t_skip * 1e3:6.0f} GB/s)   residual {t_res:6.1f}   bare {t_bare:6.1f}   x.sum() {t_copy:6.1f}', flush=True)
