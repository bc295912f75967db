"""Stream-K hand-off cost: a few layer shapes x worker counts, pure kernel durations via rocprofv3 kernel trace.
Run on the GPU box:  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sk -o sk -- python tools/bench_sk.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import hipops

SHAPES = [(512, 32, 4, 1, 0), (512, 512, 4, 3, 0), (512, 512, 32, 3, 0), (512, 512, 32, 3, 1), (512, 512, 64, 3, 0), (256, 128, 128, 3, 1)]
for i, o, r, ks, tr in SHAPES:
    x = torch.randn(1, i, r, r, device='cuda')
    wk = hipops.pack_conv_weight(torch.randn(o, i, ks, ks, device='cuda'))
    for _ in range(3):
        hipops.conv2d_mfma(x, wk, ksize=ks, transposed=bool(tr))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        hipops.conv2d_mfma(x, wk, ksize=ks, transposed=bool(tr))
    e1.record()
    torch.cuda.synchronize()
    print(f'I{i} O{o} {r}x{r} k{ks} tr{tr}: {e0.elapsed_time(e1) / n * 1e3:8.1f} us per call (events, back to back)', flush=True)
