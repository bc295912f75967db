"""N > 1 path on CPU: two gloo ranks shard a batch of frames, all-gather the images, and must reproduce the
single-process batch (including the batch-global depth range)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from invertavatar_amd import frame_parallel, synthetic


def test_shard_ranges_cover_the_batch():
    for n in (1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            spans = [frame_parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def _cams(frames):
    """Orbit cameras pushed to different distances so that the batch mean |origin| really depends on the batch."""
    c = synthetic.camera_labels(frames)
    for k in range(len(frames)):
        c[k, [3, 7, 11]] *= 1.0 + 0.03 * k
    return c


def _worker(rank, world, port, frames, nrr, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
    g = TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)
    synthetic.fill_parameters(g)
    with torch.no_grad():
        ws = g.mapping(synthetic.latent(0, 1), synthetic.conditioning_camera(), truncation_psi=0.7, truncation_cutoff=14)
        c, uv = _cams(frames), synthetic.uv_conditions(frames)
        jit = synthetic.jitter(frames, nrr * nrr)
        full = frame_parallel.render_sharded(g, ws, c, {'uvcoords_image': uv}, rank, world, jitter=jit,
                                             neural_rendering_resolution=nrr, noise_mode='const', evaluation=True)
    if rank == 0:
        torch.save(full, tmp)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharding_matches_single_process(tmp_path):
    frames, nrr = [0, 30, 60, 90], 16     # the orbit changes the camera distance per frame -> dist is truly batch-global
    out_file = str(tmp_path / 'gathered.pt')
    mp.spawn(_worker, args=(2, 29611, frames, nrr, out_file), nprocs=2, join=True)
    gathered = torch.load(out_file)
    from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
    g = TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)
    synthetic.fill_parameters(g)
    with torch.no_grad():
        ws = g.mapping(synthetic.latent(0, 1), synthetic.conditioning_camera(), truncation_psi=0.7, truncation_cutoff=14)
        c, uv = _cams(frames), synthetic.uv_conditions(frames)
        ref = g.synthesis(ws.expand(4, -1, -1), c, {'uvcoords_image': uv}, jitter=synthetic.jitter(frames, nrr * nrr),
                          neural_rendering_resolution=nrr, noise_mode='const', evaluation=True)['image']
    assert gathered.shape == ref.shape == (4, 3, 512, 512)
    assert (gathered - ref).abs().max().item() <= 2e-5
    # without the global dist the shards would disagree with the batch: make sure the test can see that
    d_all = frame_parallel.global_ray_dist(c).item()
    d_half = frame_parallel.global_ray_dist(c[:2]).item()
    assert abs(d_all - d_half) > 1e-2


def test_fewer_frames_than_ranks_is_refused_on_every_rank():
    import pytest
    from invertavatar_amd import frame_parallel
    c = synthetic.camera_labels([0, 1])
    for rank in range(4):
        with pytest.raises(ValueError, match='cannot be sharded'):
            frame_parallel.render_sharded(None, None, c, {'uvcoords_image': None}, rank=rank, world_size=4)
