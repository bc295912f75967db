"""bench.py's N > 1 code path (RCCL-free rehearsal): two ranks over gloo on CPU tensors run the SAME make_step / timed / all-gather /
JSON code the driver's 8-GPU run executes.  Small width, nrr 32: plumbing, not a measurement."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def run_bench(port, *flags, nproc=2, timeout=900):
    """port = None: plain `python bench.py --gpus N ...` (no launcher, WORLD_SIZE unset): bench.py starts its ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '2'
    launcher = [] if port is None else ['-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
                                        '--master-port', str(port)]
    cmd = [sys.executable, *launcher, os.path.join(REPO, 'bench.py'), '--gpus', str(nproc), *flags]
    r = subprocess.run(cmd, cwd=REPO, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


CPU = ('--device', 'cpu', '--dist-backend', 'gloo', '--width', 'small', '--nrr', '32', '--steps', '1', '--warmup', '0')


def test_bench_two_ranks_reenact_workload_over_gloo():
    out = run_bench(29631, *CPU, '--frames-per-rank', '1')
    assert out['n_gpus'] == 2 and out['steps'] == 1 and out['scaling'] == 'weak' and out['value'] > 0
    assert out['config']['global_batch'] == 2 and 'all_gather' in out['config']['collective']
    assert out['value'] == pytest.approx(2 * 1 / (out['ms_per_step'] * 1e-3), rel=1e-3, abs=6e-4)        # whole-job frames / max-over-ranks time (3 printed decimals)


def test_bench_two_ranks_drive_workload_over_gloo():
    out = run_bench(29633, *CPU, '--frames-per-rank', '2', '--workload', 'drive', '--drive-frames', '8')
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 4 and out['value'] > 0
    assert 'configs[4]' in out['config']['workload'] and out['config']['identity_features'] == 'backbone'


def test_bench_without_a_launcher_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` (the shape of the driver's N = 1 command) must run two ranks, not one."""
    out = run_bench(None, *CPU, '--frames-per-rank', '1')
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 2 and out['config']['parallelism'] == 'frame-sharded dp2'
