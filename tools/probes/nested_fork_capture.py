"""Which multi-stream patterns does stream capture on this HIP runtime survive?  Each variant runs in its own process (a failure is a
crash of hipStreamEndCapture, not an exception).
  direct      : origin forks s1 and s2, both join the origin
  nested      : origin -> s1 -> s2 (s2 forks from s1), s2 joins s1, s1 joins the origin
  nested_dj   : nested, and the origin also waits for s2 directly
  fork_s1_join_origin : s2 forks from s1, joins the origin only
  cross       : origin forks s1 and s2; s1 then waits for s2 (a dependency between two capturing streams), both join the origin
  prefork     : origin forks s1 and s2 FIRST; later s2 waits for s1 (what was a nested fork), s1 waits for s2, both join the origin"""
import subprocess
import sys

VARIANTS = ('direct', 'nested', 'nested_dj', 'fork_s1_join_origin', 'cross', 'prefork')

if len(sys.argv) > 1:
    import torch
    v = sys.argv[1]
    x = torch.ones(1 << 20, device='cuda')
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        a = x * 2
        if v in ('direct', 'cross', 'prefork'):
            s1.wait_stream(main)
            s2.wait_stream(main)
            with torch.cuda.stream(s1):
                b = a + 1
            if v == 'prefork':
                s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                c = (b if v == 'prefork' else a) * 3
            if v != 'direct':
                s1.wait_stream(s2)
            with torch.cuda.stream(s1):
                d = (c + b) if v != 'direct' else b * 4
            main.wait_stream(s1)
            main.wait_stream(s2)
        else:
            s1.wait_stream(main)
            with torch.cuda.stream(s1):
                b = a + 1
                s2.wait_stream(s1)
                with torch.cuda.stream(s2):
                    c = b * 3
                if v != 'fork_s1_join_origin':
                    s1.wait_stream(s2)
                    d = c + b
                else:
                    d = b * 4
            main.wait_stream(s1)
            if v in ('nested_dj', 'fork_s1_join_origin'):
                main.wait_stream(s2)
        out = d + a
    g.replay()
    torch.cuda.synchronize()
    print(v, 'ok', float(out[0]))
else:
    for v in VARIANTS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
        print(f'{v:22s} rc {r.returncode:4d}  ', ([l for l in r.stdout.strip().splitlines() if ' ok ' in l] or ['(crashed)'])[-1])
