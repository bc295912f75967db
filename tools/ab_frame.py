"""Same-box A/B of the headline step (and the 8-frame call) under environment switches: each configuration runs in its own process
(the switches are read at import / launch), alternating, so box-to-box variance cancels.
    python tools/ab_frame.py "IA_SMALL_CONV=0" "IA_SMALL_CONV=1" ...      (configurations; "default" = no switch)"""
import json
import os
import subprocess
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
configs = sys.argv[1:] or ['default']
rounds = int(os.environ.get('AB_ROUNDS', 2))
extra = os.environ.get('AB_BENCH_FLAGS', '').split()
res = {c: [] for c in configs}
for r in range(rounds):
    for c in configs:
        env = dict(os.environ)
        sets = []
        if c != 'default':
            for kv in c.split(','):
                k, v = kv.split('=')
                if '.' in k:          # module-level switch of invertavatar_amd: bench.py --set module.ATTR=value
                    sets += ['--set', kv]
                else:
                    env[k] = v
        out = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--no-extra', '--no-roofline', '--no-cpu-baseline', '--steps', '60',
                              '--warmup', '10', *sets, *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
        res[c].append(json.loads(line[0])['value'] if line else None)
        print(f'round {r} {c}: {res[c][-1]}', flush=True)
print(json.dumps(res))
