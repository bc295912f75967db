"""Compact register / spill / LDS table for the kernels of one HIP source:  python tools/kernel_resources.py conv_mfma"""
import os
import re
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
src = os.path.join(root, 'invertavatar_amd', 'csrc', sys.argv[1] + '.hip')
cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-I' + os.path.join(root, 'include'),
       '-I' + os.path.dirname(src), '-c', src, '-o', '/tmp/_res.o', '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, capture_output=True, text=True, cwd='/tmp').stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r':\d+:\d+: remark: +(.*?) \[-Rpass', line) or re.search(r':\d+:\d+: +(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:') or t.startswith('Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
        rows.append(cur)
    elif ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip()
    name = name.replace('(anonymous namespace)::', '').split('(')[0][:64]
    g = lambda k: r.get(k, '?')
    print(f"{name:64s} vgpr {g('VGPRs'):>4s} agpr {g('AGPRs'):>3s} spill {g('VGPRs Spill'):>3s} scratch {g('ScratchSize [bytes/lane]'):>4s} "
          f"occ {g('Occupancy [waves/SIMD]'):>2s} lds {g('LDS Size [bytes/block]')}")
