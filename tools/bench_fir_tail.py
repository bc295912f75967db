"""ia_fir_tail_split (FIR + noise + bias + lrelu tail of an up-sampling layer, result in split format) on the frame's shapes, timed alone."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from invertavatar_amd import hipops
from invertavatar_amd.torch_utils.ops import upfirdn2d

SHAPES = [(512, 8), (512, 16), (512, 32), (512, 64), (256, 128), (128, 256), (256, 256), (128, 512)]      # (channels, output resolution)


def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


f = upfirdn2d.setup_filter([1, 3, 3, 1]).cuda()
for c, r in SHAPES:
    t = torch.randn(1, c, r + 1, r + 1, device='cuda')
    noise, bias, sn = torch.randn(r * r, device='cuda'), torch.randn(c, device='cuda'), torch.rand(1, c, device='cuda') + 0.5
    ns = torch.full((1,), 0.3, device='cuda')
    us = bench(lambda: hipops.fir_tail_split(t, f, noise, ns, bias, styles_next=sn, out_hw=(r, r), pad0=(1, 1), fir_gain=4.0, act='lrelu',
                                             act_gain=2 ** 0.5, want_f32=False))
    nbytes = 4.0 * (t.numel() + c * r * r)
    print(f'C={c:4d} res={r:4d}  {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s', flush=True)
