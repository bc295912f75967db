"""Builds libia_hip.so (the C-ABI HIP backend) in-tree with one hipcc invocation per source.

Cross-compiles for gfx950 without a GPU.  Objects are cached by source hash so that a
rebuild after touching one kernel takes seconds.  Usage: ``python -m invertavatar_amd.build``.
"""
import hashlib
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB_PATH = os.path.join(CSRC, 'libia_hip.so')
ARCH = 'gfx950'
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-ffp-contract=off', '-Wall', '-Wno-unused-function']


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: libia_hip.so cannot be built (ROCm toolchain required)')
    return exe


def _digest(paths):
    h = hashlib.sha1(' '.join(FLAGS).encode())
    for p in sorted(paths):
        with open(p, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def headers():
    inc = os.path.join(os.path.dirname(CSRC), '..', 'include')
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith('.h')]
    return sorted(hs)


def source_digest():
    """Digest of every kernel source + header + the compile flags: stamps measurement files (profiles/) with the code they measured."""
    return _digest(sources() + headers())


def build(verbose=False, force=False):
    """Compile every csrc/*.hip for gfx950 and link libia_hip.so.  Returns its path."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, 'build')
    os.makedirs(objdir, exist_ok=True)
    hdrs = headers()
    objs, rebuilt = [], False
    procs = []
    for src in sources():
        tag = _digest([src] + hdrs)
        obj = os.path.join(objdir, f'{os.path.basename(src)[:-4]}.{tag}.o')
        objs.append(obj)
        if force or not os.path.exists(obj):
            for stale in os.listdir(objdir):
                if stale.startswith(os.path.basename(src)[:-4] + '.') and stale.endswith('.o'):
                    os.remove(os.path.join(objdir, stale))
            cmd = [hipcc, *FLAGS, '-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
            rebuilt = True
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out.decode()}')
        if verbose and out:
            print(out.decode())
    if rebuilt or force or not os.path.exists(LIB_PATH):
        cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', *objs, '-o', LIB_PATH]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout.decode()}')
    return LIB_PATH


if __name__ == '__main__':
    print(build(verbose=True, force='--force' in sys.argv))
