"""Every fp16-pair (split-format) convolution shape of a BASELINE frame, timed alone: ia_conv2d_mfma_sx per layer, optionally swept over
the DMA ring depth (env IA_RING_STAGES, read by the library at every launch).  Usage: python tools/bench_conv_layers.py [stages ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import hipops

# (in, out, input resolution, transposed, launches per frame)
LAYERS = [(512, 512, 32, 1, 3), (512, 512, 32, 0, 3), (512, 512, 64, 0, 3), (512, 256, 64, 1, 3), (256, 256, 128, 0, 3), (256, 128, 128, 1, 3),
          (128, 128, 256, 0, 3), (32, 256, 128, 1, 1), (256, 256, 256, 0, 1), (256, 128, 256, 1, 1), (128, 128, 512, 0, 1)]


def bench(fn, n=20):
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


def main():
    sweeps = sys.argv[1:] or ['default']
    batch = int(os.environ.get('BENCH_B', 1))
    rows, total = [], {s: 0.0 for s in sweeps}
    for i, o, r, tr, per_frame in LAYERS:
        x = torch.randn(batch, i, r, r, device='cuda')
        st = torch.rand(batch, i, device='cuda') + 0.5
        wk = hipops.pack_conv_weight_split(torch.randn(o, i, 3, 3, device='cuda'))
        xs = hipops.act_split(x, st)
        # BENCH_EPI=1: with the epilogue terms the layers carry in a frame (demodulation; stride-1: + noise, bias, lrelu, split output)
        kw = {}
        if os.environ.get('BENCH_EPI') == '1':
            kw = dict(demod=torch.rand(batch, o, device='cuda') + 0.5)
            if not tr:
                kw.update(noise=torch.randn(r * r, device='cuda'), noise_strength=torch.full((1,), 0.3, device='cuda'), bias=torch.randn(o, device='cuda'),
                          act='lrelu', gain=2 ** 0.5, want_f32=False, styles_next=torch.rand(batch, o, device='cuda') + 0.5)
        fl = 2.0 * batch * r * r * 9 * i * o
        line = f'I={i:4d} O={o:4d} res={r:4d} tr={tr}'
        ref = None
        for sname in sweeps:
            # a sweep entry is "default" or '+'-separated ENV=VALUE pairs, e.g. IA_RING_STAGES=2+IA_DMA_SPREAD=1
            for k in ('IA_RING_STAGES', 'IA_DMA_SPREAD', 'IA_XCD_BANDS', 'IA_SX_TILE', 'IA_SX_WHOLE', 'IA_NO_NARROW_TILES'):
                os.environ.pop(k, None)
            if sname != 'default':
                for kv in sname.split('+'):
                    k, v = kv.split('=')
                    os.environ[k] = v
            try:
                y = hipops.conv2d_mfma_sx(xs, wk, transposed=bool(tr), **kw)
                y = y.float() if isinstance(y, hipops.SplitAct) else y
            except RuntimeError as err:      # (a forced tile family that does not cover this shape)
                line += f' | {"n/a":>7s}    ' + str(err)[:40]
                total[sname] += float('nan')
                continue
            if ref is None:
                ref = y
            dev = float((y - ref).abs().max() / ref.abs().max())
            us = bench(lambda: hipops.conv2d_mfma_sx(xs, wk, transposed=bool(tr), **kw))
            total[sname] += us * per_frame
            line += f' | {us:7.1f} us {fl / us / 1e6:6.1f} TF {3 * fl / us / 1e6 / 2500:.3f} d={dev:.1e}'
        print(line, flush=True)
    print('sweeps: ' + ' | '.join(sweeps))
    print('per frame (us): ' + ', '.join(f'{k}: {v:.0f}' for k, v in total.items()))


if __name__ == '__main__':
    main()
