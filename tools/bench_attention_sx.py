"""Attention of the large token grids: hipops.attention_sx (fp16-pair GEMMs, three launches + splits) against the ATen sequence of
Attention.forward (two batched rocBLAS GEMMs, scale, softmax, permutes):  python tools/bench_attention_sx.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from invertavatar_amd import hipops


def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for n in (1024, 4096):
    c, heads = 1024, 4
    hd, scale = c // heads, (c // heads) ** -0.5
    q, kv = torch.randn(1, n, c, device='cuda'), torch.randn(1, n, 2 * c, device='cuda')

    def aten():
        qq = q.reshape(1, n, heads, hd).permute(0, 2, 1, 3)
        k, v = kv.reshape(1, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
        return (((qq @ k.transpose(-2, -1)) * scale).softmax(dim=-1) @ v).transpose(1, 2).reshape(1, n, c)
    t_a, t_s = bench(aten), bench(lambda: hipops.attention_sx(q, kv, heads, scale))
    err = (hipops.attention_sx(q, kv, heads, scale) - aten()).abs().max().item()
    print(f'N = M = {n}: ATen {t_a:7.1f} us   attention_sx {t_s:7.1f} us   max |d| {err:.1e}', flush=True)
