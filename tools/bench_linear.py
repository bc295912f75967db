"""The one-shot flow's linear layers: ia_tokens_split + ia_linear_sx against F.linear (rocBLAS fp32) on the shapes of the transformer blocks
(1 024 dims, mlp_ratio 2; 64^2 .. 8^2 token grids):  python tools/bench_linear.py     (IA_LINEAR_TILE=1|2|3 forces a tile form)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from invertavatar_amd import hipops

SHAPES = [(4096, 1024, 1024), (4096, 1024, 2048), (4096, 2048, 1024), (1024, 1024, 1024), (1024, 1024, 2048), (1024, 2048, 1024),
          (256, 1024, 1024), (256, 1024, 2048), (256, 2048, 1024), (64, 1024, 2048), (64, 2048, 1024)]


def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for m, k, n in SHAPES:
    x = torch.randn(1, m, k, device='cuda')
    w = torch.randn(n, k, device='cuda') * 0.02
    b = torch.randn(n, device='cuda')
    ws = hipops.pack_linear_weight_split(w)
    xs = hipops.tokens_split(x)
    t_lib = bench(lambda: torch.nn.functional.linear(x, w, b))
    t_split = bench(lambda: hipops.tokens_split(x))
    t_lin = bench(lambda: hipops.linear_sx(xs, ws, b))
    if '--graph' in sys.argv:      # kernel time without the host's launch floor: the call replayed as a hipGraph
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            hipops.linear_sx(xs, ws, b)
        t_lin = bench(g.replay)
        gl = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gl):
            torch.nn.functional.linear(x, w, b)
        t_lib = bench(gl.replay)
    err = (hipops.linear_sx(xs, ws, b).double() - torch.nn.functional.linear(x.double(), w.double(), b.double())).abs().max().item()
    gf = 2.0 * m * k * n
    print(f'M={m:5d} K={k:5d} N={n:5d}   F.linear {t_lib:7.1f} us ({gf / t_lib / 1e6:6.1f} TF)   tokens_split {t_split:6.1f} us   linear_sx {t_lin:7.1f} us '
          f'({gf / t_lin / 1e6:6.1f} TF fp32-equivalent)   max |d| vs fp64 {err:.1e}', flush=True)
