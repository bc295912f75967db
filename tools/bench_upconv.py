"""The up-sampling (transposed) layers of a BASELINE frame, four-phase tile (ia_conv2d_mfma_sx transposed) against the row-phase
form (ia_upconv2d_rows_sx):  python tools/bench_upconv.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import hipops
from bench_conv_layers import bench

LAYERS = [(512, 512, 16, 3), (512, 512, 32, 3), (512, 256, 64, 3), (256, 128, 128, 3), (256, 128, 256, 1)]
batch = int(os.environ.get('BENCH_B', 1))
tot = [0.0, 0.0]
for i, o, r, per_frame in LAYERS:
    x = torch.randn(batch, i, r, r, device='cuda')
    st = torch.rand(batch, i, device='cuda') + 0.5
    wk = hipops.pack_conv_weight_split(torch.randn(o, i, 3, 3, device='cuda'))
    xs = hipops.act_split(x, st)
    d = torch.rand(batch, o, device='cuda') + 0.5
    fl = 2.0 * batch * r * r * 9 * i * o
    a = hipops.conv2d_mfma_sx(xs, wk, demod=d, transposed=True)
    t4 = bench(lambda: hipops.conv2d_mfma_sx(xs, wk, demod=d, transposed=True))
    line = f'I={i:4d} O={o:4d} res={r:4d} | four-phase {t4:7.1f} us {3 * fl / t4 / 1e6 / 2500:.3f}'
    if hipops.upconv_rows_supported(batch, i, o, r, r):
        bb = hipops.upconv2d_rows_sx(xs, wk, demod=d)
        t2 = bench(lambda: hipops.upconv2d_rows_sx(xs, wk, demod=d))
        line += f' | row-phase {t2:7.1f} us {3 * fl / t2 / 1e6 / 2500:.3f}  d={float((a - bb).abs().max() / a.abs().max()):.1e}'
        tot[1] += t2 * per_frame
    else:
        tot[1] += t4 * per_frame
    tot[0] += t4 * per_frame
    print(line, flush=True)
print(f'per frame (us): four-phase {tot[0]:.0f}, row-phase where supported {tot[1]:.0f}')
