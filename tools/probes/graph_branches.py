"""Probe: do parallel branches of a captured hipGraph run concurrently?  B branches x N dependent ~10 us spin kernels, forked from
and joined to the capture stream; reports the wall time of a replay against the serial (B*N*t) and parallel (N*t) bounds."""
import sys
import time

import torch

B, N, CYC = int(sys.argv[1]) if len(sys.argv) > 1 else 3, 40, 20000
dev = torch.device('cuda')
streams = [torch.cuda.Stream() for _ in range(B)]


def body():
    main = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(main)
        with torch.cuda.stream(s):
            for _ in range(N):
                torch.cuda._sleep(CYC)
    for s in streams:
        main.wait_stream(s)


def timed(fn, reps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    one = timed(lambda: [torch.cuda._sleep(CYC) for _ in range(N)], 5)
    print(f'one branch of {N} kernels, eager: {one:.0f} us')
    print(f'{B} branches eager multi-stream: {timed(body):.0f} us')
    body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        body()
    print(f'{B} branches graph replay: {timed(g.replay):.0f} us   (serial bound {B * one:.0f}, parallel bound {one:.0f})')
