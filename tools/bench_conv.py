"""Micro-benchmark of ia_conv2d_mfma on the layer shapes of the BASELINE model (TFLOP/s vs the 157.3 fp32 MFMA peak)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from invertavatar_amd import hipops

SHAPES = [  # I, O, res(in), transposed
    (512, 512, 4, 0), (512, 512, 8, 0), (512, 512, 16, 0), (512, 512, 32, 0), (512, 512, 64, 0), (256, 256, 128, 0), (128, 128, 256, 0),
    (512, 512, 4, 1), (512, 512, 16, 1), (512, 512, 32, 1), (512, 256, 64, 1), (256, 128, 128, 1),
    (32, 256, 128, 1), (256, 256, 256, 0), (256, 128, 256, 1), (128, 128, 512, 0),
]
for i, o, r, tr in SHAPES:
    x = torch.randn(1, i, r, r, device='cuda')
    wk = hipops.pack_conv_weight(torch.randn(o, i, 3, 3, device='cuda'))
    fn = lambda: hipops.conv2d_mfma(x, wk, ksize=3, transposed=bool(tr))
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * r * r * 9 * i * o
    print(f'I={i:4d} O={o:4d} res={r:4d} tr={tr}  {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s', flush=True)
