"""fp16-operand form vs fp32 form of the SR head's layers."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from invertavatar_amd import hipops
for i, o, r, tr in [(32, 256, 128, 1), (256, 256, 256, 0), (256, 128, 256, 1), (128, 128, 512, 0), (512, 512, 64, 0), (256, 128, 128, 1)]:
    x = torch.randn(1, i, r, r, device='cuda')
    w = torch.randn(o, i, 3, 3, device='cuda')
    for name, wk in (('f32', hipops.pack_conv_weight(w)), ('f16', hipops.pack_conv_weight_h(w)), ('f16x3', hipops.pack_conv_weight_split(w))):
        fn = lambda: hipops.conv2d_mfma(x, wk, ksize=3, transposed=bool(tr))
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f'I={i:4d} O={o:4d} res={r:4d} tr={tr} {name}: {ms*1e3:8.1f} us  {2.0*r*r*9*i*o/ms/1e9:7.1f} TFLOP/s', flush=True)
