"""BASELINE configs[4] on the device: few-shot ConvGRU inversion of 8 source frames (the script's flow) -> identity features -> drive
frames through synthesis_withTexture in calls of 8 with the SR head in its deployed fp16 precision; and the N > 1 path of bench.py
rehearsed with two ranks sharing the one GPU over gloo."""
import os

import pytest
import torch

from invertavatar_amd import synthetic
from conftest import max_abs

pytestmark = pytest.mark.gpu

TOL_BATCH = 2e-5           # B = 8 call vs the same frames one call each (HIP vs HIP: summation order of batch-shaped launches only)
TOL_RGB_FP16_SR = 4e-3     # fp16 SR head vs the fp32 head on the same features (tests/test_generator_gpu.py has the derivation)


@pytest.fixture(scope='module')
def inverted():
    """Full-width generator with sr_num_fp16_res = 4 + the inversion network; features from the harness' default (script) flow."""
    from invertavatar_amd import eval_seq
    from invertavatar_amd.encoder_inversion.models.uvnet import inversionNet
    from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
    from encoder_common import source_batch
    g = TriPlaneGenerator(**synthetic.generator_kwargs('full', sr_num_fp16_res=4)).eval().requires_grad_(False)
    synthetic.fill_parameters(g)
    net = inversionNet(generator=g, encoding_triplane=True, encoding_texture=True).requires_grad_(False)
    synthetic.fill_encoder_parameters(net)
    net = eval_seq.set_eval_seq_modes(net.cuda())
    g.neural_rendering_resolution = 32                       # (the inversion's own renders; the drive frames below use 128)
    src = source_batch('cuda')
    ws, res, _ = eval_seq.few_shot_inversion(net, src['image'], src['uv'], src['c'], src['uvcoords'])
    return net, ws, res


def _drive_inputs(frames, nrr):
    return (synthetic.camera_labels(frames).cuda(), synthetic.uv_conditions(frames).cuda(), synthetic.jitter(frames, nrr * nrr).squeeze(-1).cuda())


def test_config4_drive_block_of_eight_fp16_sr_on_inverted_features(inverted):
    from invertavatar_amd import eval_seq
    from invertavatar_amd.frame_parallel import per_frame_ray_dist
    from invertavatar_amd.graphed import GraphedDrive
    from invertavatar_amd.training import networks_stylegan2 as sg2
    net, ws, res = inverted
    nrr, frames = 128, [40, 47, 61, 90, 120, 155, 200, 233]
    c, uv, jit = _drive_inputs(frames, nrr)
    saved, sg2.FP16_BLOCKS_COMPUTE_FP32 = sg2.FP16_BLOCKS_COMPUTE_FP32, False
    try:
        with torch.no_grad():
            one, _ = eval_seq.drive_sequence(net, ws, res, c, uv, jitter=jit, batch=1, neural_rendering_resolution=nrr, graphed=False)   # the script: B = 1, eager
            blk, _ = eval_seq.drive_sequence(net, ws, res, c, uv, jitter=jit, batch=8, neural_rendering_resolution=nrr)     # one call of 8
            dflt, _ = eval_seq.drive_sequence(net, ws, res, c, uv, jitter=jit, neural_rendering_resolution=nrr)             # the harness default: captured calls of 8
            dflt9, _ = eval_seq.drive_sequence(net, ws, res, torch.cat([c, c[:1]]), torch.cat([uv, uv[:1]]), jitter=torch.cat([jit, jit[:1]]),
                                               neural_rendering_resolution=nrr)                                            # 8 + a one-frame call
            graphed = GraphedDrive(net.generator, ws, res['texture'], res['static'], batch=8, neural_rendering_resolution=nrr, ray_dist_elems=8)
            rep = graphed(c, uv, jit, per_frame_ray_dist(c))['image'].clone()
            sg2.FP16_BLOCKS_COMPUTE_FP32 = True                                                                               # fp32 head, same features
            one32, _ = eval_seq.drive_sequence(net, ws, res, c, uv, jitter=jit, batch=1, neural_rendering_resolution=nrr, graphed=False)
            f32, _ = eval_seq.drive_sequence(net, ws, res, c, uv, jitter=jit, batch=8, neural_rendering_resolution=nrr)
    finally:
        sg2.FP16_BLOCKS_COMPUTE_FP32 = saved
    assert blk.shape == (8, 3, 512, 512) and torch.isfinite(blk).all()
    d_batch32, d_batch16, d_graph, d_f32 = max_abs(f32, one32), max_abs(blk, one), max_abs(rep, blk), max_abs(blk, f32)
    print(f'configs[4] drive block: B=8 vs 8 x B=1 {d_batch32:.2e} (fp32 head) / {d_batch16:.2e} (fp16 head), graph vs eager {d_graph:.2e}, '
          f'fp16 SR vs fp32 SR {d_f32:.2e}')
    # Batching changes the summation order of the batch-shaped launches only: 2e-5 with the fp32 head.  With the fp16 head those
    # last-bit differences meet the fp16 rounding of every stored activation (a value next to a rounding boundary lands on the other
    # side: 2^-11 relative), so the batched and the one-frame calls agree to the fp16 mode's own tolerance, not to 2e-5.
    assert d_batch32 <= TOL_BATCH
    assert d_batch16 <= TOL_RGB_FP16_SR
    assert d_graph == 0.0                                    # the captured call replays the eager call bit for bit
    # eval_seq.drive_sequence's default: captured calls of 8 with per-frame depth ranges = the eager call of 8, bit for
    # bit; a ninth frame goes through the captured one-frame call = the script's own call
    assert max_abs(dflt, blk) == 0.0 and max_abs(dflt9[:8], blk) == 0.0 and max_abs(dflt9[8], one[0]) == 0.0
    assert 0.0 < d_f32 <= TOL_RGB_FP16_SR                    # (> 0: the fp16 mode really ran)
    assert max_abs(blk[0], blk[1]) > 1e-2                    # different drive frames differ


def test_synthesis_with_random_noise_mode_runs_on_the_device_path():
    """noise_mode='random' (the default of G(z, c, v)) must not hand a SplitAct to a layer that takes the
    random-noise branch.  With every noise_strength at zero the random-noise route equals the const-noise route."""
    from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
    g = TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)
    synthetic.fill_parameters(g)
    for name, p in g.named_parameters():
        if name.endswith('noise_strength'):
            p.zero_()
    g = g.cuda()
    frames, nrr = [3, 77], 32
    c, uv, jit = _drive_inputs(frames, nrr)
    with torch.no_grad():
        ws = g.mapping(synthetic.latent(5, 2).cuda(), synthetic.conditioning_camera().cuda().expand(2, -1), truncation_psi=0.7, truncation_cutoff=14)
        kw = dict(neural_rendering_resolution=nrr, evaluation=False, jitter=jit, return_featmap=True)
        a = g.synthesis(ws, c, {'uvcoords_image': uv}, noise_mode='random', **kw)['triplane']       # (the renderer draws its own samples
        b = g.synthesis(ws, c, {'uvcoords_image': uv}, noise_mode='const', **kw)['triplane']        # without evaluation: compare the planes)
        out = g(synthetic.latent(5, 2).cuda(), c, {'uvcoords_image': uv}, neural_rendering_resolution=nrr)      # defaults: noise_mode random
    assert torch.isfinite(out['image']).all() and out['image'].shape == (2, 3, 512, 512)
    assert max_abs(a, b) <= 2e-4, max_abs(a, b)


@pytest.mark.parametrize('workload,extra', [('reenact', ()), ('drive', ('--features', 'backbone', '--drive-frames', '8'))])
def test_bench_n_gt_1_path_two_ranks_on_one_gpu(workload, extra):
    """bench.py --gpus 2 with both ranks on cuda:0 and the collectives over gloo: device tensors, captured graphs with the
    batch-global / per-frame ray_dist input, the all-gather and the max-over-ranks timing all execute (RCCL itself is the driver's)."""
    from test_bench_cpu import run_bench
    # reenact: plain `python bench.py --gpus 2` (bench.py starts its own ranks); drive: under torch.distributed.run, as the driver's N > 1 command
    out = run_bench(None if workload == 'reenact' else 29643, '--dist-backend', 'gloo', '--width', 'full', '--steps', '2', '--warmup', '1',
                    '--frames-per-rank', '2', '--workload', workload, *extra, timeout=1500)
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 4 and out['value'] > 0
    assert out['config']['launch'].startswith('hipGraph replay'), out['config']['launch']


def _run_bench_one_rank(*flags, timeout=1500):
    import json
    import os
    import subprocess
    import sys
    from test_bench_cpu import REPO
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR')}
    env['MASTER_PORT'] = '29647'
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '1', *flags], cwd=REPO, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('workload,extra', [('reenact', ()), ('drive', ('--features', 'backbone', '--drive-frames', '16'))])
def test_bench_force_dist_runs_rccl_at_world_size_one(workload, extra):
    """RCCL itself executes on the one-GPU box.  `bench.py --gpus 1 --force-dist` brings up
    init_process_group('nccl', world_size=1, device_id=cuda:0) with HSA_ENABLE_IPC_MODE_LEGACY=0 and ends every timed step in the
    all_gather_into_tensor of the N > 1 path (+ its barriers and the max-over-ranks all-reduce)."""
    out = _run_bench_one_rank('--force-dist', '--steps', '3', '--warmup', '1', '--frames-per-rank', '2', '--workload', workload, *extra)
    assert out['n_gpus'] == 1 and out['value'] > 0 and 'all_gather' in out['config']['collective']
    assert 'nccl' in out['config'].get('force_dist', '') or workload == 'drive'
    assert out['config']['launch'].startswith('hipGraph replay'), out['config']['launch']


@pytest.mark.timeout(900)
def test_render_sharded_over_rccl_world_size_one_equals_the_plain_call():
    """frame_parallel.render_sharded with a live RCCL process group (world size 1): all_gather_blocks goes through
    all_gather_into_tensor on the device communicator and returns the frames of the plain synthesis call, bit for bit."""
    import os
    from invertavatar_amd import frame_parallel
    from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', '29649'
    g = synthetic.fill_parameters(TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)).cuda()
    frames, nrr = [3, 77, 140], 32
    c, uv, jit = _drive_inputs(frames, nrr)
    torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        with torch.no_grad():
            ws = g.mapping(synthetic.latent(5, 1).cuda(), synthetic.conditioning_camera().cuda(), truncation_psi=0.7, truncation_cutoff=14)
            kw = dict(neural_rendering_resolution=nrr, noise_mode='const', evaluation=True)
            ref = g.synthesis(ws.expand(3, -1, -1), c, {'uvcoords_image': uv}, jitter=jit, **kw)['image']
            # world_size 1 returns before the collective: call the collective itself on the rank's block, then the sharded entry
            got = frame_parallel.all_gather_blocks(ref.contiguous(), [3])
            assert torch.equal(got, ref)
            t = torch.ones(4, device='cuda')
            torch.distributed.all_reduce(t)
            torch.distributed.barrier()
            assert t.tolist() == [1.0] * 4
            again = frame_parallel.render_sharded(g, ws, c, {'uvcoords_image': uv}, rank=0, world_size=1, jitter=jit, **kw)
            assert torch.equal(again, ref)
    finally:
        torch.distributed.destroy_process_group()
