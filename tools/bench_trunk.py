"""IR-SE50 trunk of a UNet encoder on a 4-frame group at 256^2: residual units through ia_conv2d_mfma_sx vs the library route."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd.encoder_inversion.models import helpers, trunk_hip

torch.manual_seed(0)
inp, body = helpers.irse50_trunk(6)
inp, body = inp.cuda().eval().requires_grad_(False), body.cuda().eval().requires_grad_(False)
x = torch.randn(4, 6, 256, 256, device='cuda')
with torch.no_grad():
    for flag in (True, False, True):
        helpers.HIP_TRUNK = flag
        h = inp(x)
        used = sum(trunk_hip.unit_supported(u, torch.empty(4, u.res_layer[1].in_channels, r, r, device='cuda'))
                   for u, r in zip(body, [256] + [128] * 3 + [64] * 4 + [32] * 14 + [16] * 3)) if flag else 0
        for _ in range(3):
            out, taps = helpers.run_trunk(body, h, (2, 6, 20, 21))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(5):
            out, taps = helpers.run_trunk(body, h, (2, 6, 20, 21))
        e1.record()
        torch.cuda.synchronize()
        print(f'HIP_TRUNK={flag}: {e0.elapsed_time(e1) / 5:.2f} ms GPU, {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms host, units on HIP: {used}/24', flush=True)
        if flag:
            ref = out.clone()
        else:
            print('max relative deviation HIP vs library:', ((ref - out).abs().max() / out.abs().max()).item())
    # per unit
    helpers.HIP_TRUNK = True
    h = inp(x)
    cur = h
    for i, u in enumerate(body):
        for route in ('hip', 'lib'):
            if route == 'hip' and not trunk_hip.unit_supported(u, cur):
                continue
            fn = (lambda: trunk_hip.unit_forward(u, cur)) if route == 'hip' else (lambda: u(cur))
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                o = fn()
            e1.record()
            torch.cuda.synchronize()
            print(f'unit {i:2d} {tuple(cur.shape)} -> {tuple(o.shape)} {route}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us', flush=True)
        cur = u(cur)

    # the same trunk pass as a captured graph: is the eager figure host time?
    for flag in (True, False):
        helpers.HIP_TRUNK = flag
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                helpers.run_trunk(body, h, (2, 6, 20, 21))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out, taps = helpers.run_trunk(body, h, (2, 6, 20, 21))
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        print(f'captured graph, HIP_TRUNK={flag}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per trunk pass', flush=True)
