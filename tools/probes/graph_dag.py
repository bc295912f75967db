"""Probe: how does the hipGraph executor schedule a DAG shaped like the frame?  Spin kernels stand in for the networks:
mouth (1 x 200 us), face head (35 x 10), texture (60 x 15, taps after 20/35/50), static (55 x 25, taps after 15/25/35),
raster level k (waits for tap k of both + mouth; 50 us each), face tail (waits head + level k; 3 x 8 x 12 us), then 1400 us of
render + SR.  Prints the replay wall time of several ways of expressing the same dependencies."""
import sys
import time

import torch

US = 2100        # _sleep cycles per microsecond (approximate; calibrated below)


def spin(us):
    torch.cuda._sleep(int(us * US))


def chain(n, us, taps=(), events=None, stream=None):
    for i in range(n):
        spin(us)
        if i + 1 in taps:
            ev = torch.cuda.Event()
            ev.record(stream)
            events.append(ev)


def frame(variant):
    main = torch.cuda.current_stream()
    S = {k: streams[k] for k in ('mouth', 'head', 'tex', 'sta', 'raster')}
    if variant == 'aux_one':
        S['head'] = S['mouth']
    if variant == 'sta_main':
        S['sta'] = main
    if variant == 'no_raster_stream':
        S['raster'] = main
    order = ['mouth', 'head', 'tex', 'sta']
    if variant == 'sta_first':
        order = ['sta', 'tex', 'mouth', 'head']
    if variant == 'two_only':
        order = ['tex', 'sta']
    ev = {k: [] for k in order}
    done = {}
    for k in order:
        s = S[k]
        if s is not main:
            s.wait_stream(main)
        with torch.cuda.stream(s):
            if k == 'mouth':
                chain(1, 200)
            elif k == 'head':
                chain(35, 10)
            elif k == 'tex':
                chain(60, 15, (20, 35, 50), ev['tex'], s)
            else:
                chain(55, 25, (15, 25, 35), ev['sta'], s)
            e = torch.cuda.Event()
            e.record(s)
            done[k] = e
    rs = S['raster']
    if rs is not main:
        rs.wait_stream(main)
    lv = []
    with torch.cuda.stream(rs):
        if 'mouth' in done:
            rs.wait_event(done['mouth'])
        for k in range(3):
            rs.wait_event(ev['tex'][k])
            rs.wait_event(ev['sta'][k])
            spin(50)
            e = torch.cuda.Event()
            e.record(rs)
            lv.append(e)
    if 'head' in done:
        main.wait_event(done['head'])
    for k in range(3):
        main.wait_event(lv[k])
        chain(8, 12)
    main.wait_event(done['sta'])
    chain(1, 1400)
    main.wait_event(done['tex'])


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


streams = {k: torch.cuda.Stream() for k in ('mouth', 'head', 'tex', 'sta', 'raster', 'cap')}
with torch.cuda.stream(streams['cap']):
    spin(10)
    torch.cuda.synchronize()
    t = timed(lambda: spin(1000), 5)
    print(f'calibration: spin(1000) = {t:.0f} us')
    US = US * 1000 / t
    print('critical path if everything overlaps: static 1375 + level 50 + tail 96 + 1400 = 2921 us; serial sum = 200+350+900+1375+150+288+1400 = 4663 us')
    for variant in ('as_is', 'aux_one', 'sta_first', 'sta_main', 'no_raster_stream', 'two_only'):
        frame(variant)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams['cap']):
            frame(variant)
        print(f'{variant:18s} eager {timed(lambda: frame(variant), 5):7.0f} us   graph replay {timed(g.replay):7.0f} us', flush=True)
