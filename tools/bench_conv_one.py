"""One conv shape, a few launches (for PMC collection): python tools/bench_conv_one.py I O res transposed"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from invertavatar_amd import hipops
i, o, r, tr = [int(v) for v in sys.argv[1:5]]
x = torch.randn(1, i, r, r, device='cuda')
wk = hipops.pack_conv_weight(torch.randn(o, i, 3, 3, device='cuda'))
for _ in range(5):
    hipops.conv2d_mfma(x, wk, ksize=3, transposed=bool(tr))
torch.cuda.synchronize()
