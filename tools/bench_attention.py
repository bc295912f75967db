"""ia_attention vs the ATen route (two batched matmuls + softmax) at the token counts of the transformer-refined decoder stages."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import hipops


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


heads, hd = 4, 256
c = heads * hd
for n in (64, 256, 1024, 4096):
    q = torch.randn(1, n, c, device='cuda')
    kv = torch.randn(1, n, 2 * c, device='cuda')
    scale = hd ** -0.5

    def aten():
        qh = q.reshape(1, n, heads, hd).permute(0, 2, 1, 3)
        k, v = kv.reshape(1, n, 2, heads, hd).permute(2, 0, 3, 1, 4)
        return (((qh @ k.transpose(-2, -1)) * scale).softmax(dim=-1) @ v).transpose(1, 2).reshape(1, n, c)
    ok = hipops._lib.load().ia_attention_supported(hd, n, n)
    t_hip, t_aten = (bench(lambda: hipops.attention(q, kv, heads, scale)) if ok else float('nan')), bench(aten)
    fl = 4.0 * n * n * c
    print(f'N={n:5d}: ia_attention {t_hip:8.1f} us ({fl / t_hip / 1e6:6.1f} TF)   ATen {t_aten:8.1f} us', flush=True)
