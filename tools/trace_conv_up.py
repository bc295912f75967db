"""Segment timeline of ia_upconv2d_rows_sx's K loop: a build with -DIA_UP_TRACE=1 (tools/_variants/libia_up_trace.so) lets workgroup 0 stamp
s_memtime at the boundaries of its LOAD / COMPUTE segments; this prints the per-wave durations (shader cycles).
  build (CPU):  hipcc ... -DIA_UP_TRACE=1 (see tools/ablate_conv_up.sh for the link line)     run (GPU):  python tools/trace_conv_up.py I O res"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('IA_HIP_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), '_variants', 'libia_up_trace.so'))
import torch

from invertavatar_amd import hipops

i, o, r = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 128, 256)
x = torch.randn(1, i, r, r, device='cuda')
wk = hipops.pack_conv_weight_split(torch.randn(o, i, 3, 3, device='cuda'))
xs = hipops.act_split(x, torch.rand(1, i, device='cuda') + 0.5)
for _ in range(3):
    hipops.upconv2d_rows_sx(xs, wk)
torch.cuda.synchronize()
scratch = hipops._scratch_buffer(x.device, 1 << 20)
scratch.zero_()
torch.cuda.synchronize()
hipops.upconv2d_rows_sx(xs, wk)
torch.cuda.synchronize()
t = scratch[:8 * 64 * 8 * 2].view(torch.int64).reshape(8, 64, 8).cpu()      # (the edge workers' slabs start further in: E * gpe slabs)
names = ['load', 'wait b1', 'compute', 'wait b2', 'next']
for w in (0, 4, 1, 5):
    print(f'wave {w}: k-step: ' + ' '.join(f'{n:>8s}' for n in names))
    for k in range(3, 15):
        st = t[w, k]
        nxt = t[w, k + 1, 0]
        d = [int(st[1] - st[0]), int(st[2] - st[1]), int(st[3] - st[2]), int(st[4] - st[3]), int(nxt - st[4])]
        print(f'          {k:3d}   ' + ' '.join(f'{v:8d}' for v in d))
per = (t[:, 40, 0] - t[:, 4, 0]).float() / 36
print('cycles per k-step (36 k-steps), per wave:', [int(v) for v in per])
