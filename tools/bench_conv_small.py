"""The low-resolution 3x3 layers of a frame and of the inversion encoders, timed alone (ia_conv2d_mfma_sx with the epilogue terms they carry):
the in-workgroup K split of csrc/conv_small.h against a -DIA_CONV_SMALL=0 build of the same tree (stream-K tiles + fix-up launch).
Usage: python tools/bench_conv_small.py            (IA_HIP_LIB=tools/_variants/libia_nosmall.so for the other library)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import hipops

# (in, out, H, W, batch, note)
LAYERS32 = [(512, 512, 32, 32, 1, 'backbone b32.conv1'), (256, 256, 32, 32, 1, ''), (128, 128, 32, 32, 1, ''), (384, 384, 32, 32, 1, 'ConvGRU @32^2 (384)'),
            (768, 768, 32, 32, 1, 'ConvGRU ih @32^2'), (256, 256, 32, 32, 4, '4 sources')]
LAYERS = [(512, 512, 8, 8, 1, 'backbone b8.conv1'), (512, 512, 16, 16, 1, 'backbone b16.conv1'), (512, 512, 8, 8, 8, 'b8.conv1, 8 frames'),
          (512, 512, 16, 16, 8, 'b16.conv1, 8 frames'), (1024, 1024, 16, 16, 1, 'ConvGRU ih @16^2'), (1024, 512, 16, 16, 1, 'ConvGRU hh @16^2'),
          (512, 512, 16, 16, 4, 'trunk unit @16^2, 4 sources'), (256, 256, 16, 16, 1, ''), (128, 128, 12, 20, 2, 'ragged')]


def bench(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    print('library:', os.environ.get('IA_HIP_LIB', 'in-tree'))
    graph = os.environ.get('BENCH_GRAPH', '1') == '1'
    for i, o, h, w, b, note in (LAYERS32 if os.environ.get('BENCH_32') == '1' else LAYERS):
        x = torch.randn(b, i, h, w, device='cuda')
        st = torch.rand(b, i, device='cuda') + 0.5
        wt = torch.randn(o, i, 3, 3, device='cuda')
        wk = hipops.pack_conv_weight_split(wt)
        xs = hipops.act_split(x, st)
        kw = dict(demod=torch.rand(b, o, device='cuda') + 0.5, noise=torch.randn(h * w, device='cuda'), noise_strength=torch.full((1,), 0.3, device='cuda'),
                  bias=torch.randn(o, device='cuda'), act='lrelu', gain=2 ** 0.5, styles_next=torch.rand(b, o, device='cuda') + 0.5)
        fn = lambda: hipops.conv2d_mfma_sx(xs, wk, **kw)
        y, _ = fn()
        ref = torch.nn.functional.conv2d(xs.float().double(), wt.double(), padding=1) * kw['demod'].double()[:, :, None, None]
        ref = ref + (kw['noise'].double() * 0.3).view(1, 1, h, w) + kw['bias'].double()[None, :, None, None]
        ref = torch.nn.functional.leaky_relu(ref, 0.2) * 2 ** 0.5
        err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
        us = bench(fn)
        line = f'I={i:4d} O={o:4d} {h:3d}x{w:<3d} B={b}  eager {us:7.1f} us'
        if graph:      # the launch (pair) as a frame's graph replays it: 20 back-to-back launches captured, replayed
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(20):
                    fn()
            line += f'   graph {bench(g.replay, 20) / 20:7.1f} us'
        print(line + f'   rel err vs fp64 {err:.1e}   {note}', flush=True)


def main_transposed():
    print('-- transposed (up-sampling layers, four-phase form; demodulation only)')
    for i, o, r, b in [(512, 512, 8, 1), (512, 512, 16, 1), (512, 512, 8, 8), (256, 256, 16, 1)]:
        x = torch.randn(b, i, r, r, device='cuda')
        wt = torch.randn(o, i, 3, 3, device='cuda')
        wk = hipops.pack_conv_weight_split(wt)
        xs = hipops.act_split(x, torch.rand(b, i, device='cuda') + 0.5)
        d = torch.rand(b, o, device='cuda') + 0.5
        fn = lambda: hipops.conv2d_mfma_sx(xs, wk, demod=d, transposed=True)
        y = fn()
        ref = torch.nn.functional.conv_transpose2d(xs.float().double(), wt.double().transpose(0, 1), stride=2) * d.double()[:, :, None, None]
        err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
        us = bench(fn)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        print(f'I={i:4d} O={o:4d} {r:3d}x{r:<3d} B={b}  eager {us:7.1f} us   graph {bench(g.replay, 20) / 20:7.1f} us   rel err vs fp64 {err:.1e}', flush=True)


if __name__ == '__main__':
    main()
    main_transposed()
