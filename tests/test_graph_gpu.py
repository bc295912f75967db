"""The launch modes bench.py times (hipGraph replay, frames in flight) against eager calls of the same frames: replaying
a captured frame with new inputs must give the bits of an eager call -- same kernels, same order of additions."""
import pytest
import torch

from invertavatar_amd import synthetic
from invertavatar_amd.graphed import FramePipeline, GraphedSynthesis
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator

pytestmark = pytest.mark.gpu


def _setup(width, nrr, frames):
    g = TriPlaneGenerator(**synthetic.generator_kwargs(width)).eval().requires_grad_(False)
    synthetic.fill_parameters(g)
    g = g.cuda()
    with torch.no_grad():
        ws = g.mapping(synthetic.latent(0, 1).cuda(), synthetic.conditioning_camera().cuda(), truncation_psi=0.7, truncation_cutoff=14)
    cams, uvs = synthetic.camera_labels(frames).cuda(), synthetic.uv_conditions(frames).cuda()
    jits = synthetic.jitter(frames, nrr * nrr).squeeze(-1).cuda()

    def eager(i):
        with torch.no_grad():
            out = g.synthesis(ws, cams[i:i + 1], {'uvcoords_image': uvs[i:i + 1]}, neural_rendering_resolution=nrr, noise_mode='const',
                              evaluation=True, jitter=jits[i:i + 1])
        return out['image'].clone(), out['image_depth'].clone()
    return g, ws, cams, uvs, jits, eager


def test_graph_replay_is_the_eager_frame_bit_for_bit():
    frames = [0, 37, 120, 201]
    g, ws, cams, uvs, jits, eager = _setup('small', 64, frames)
    want = [eager(i) for i in range(len(frames))]
    graphed = GraphedSynthesis(g, batch=1, neural_rendering_resolution=64)
    for i in (0, 1, 2, 3, 1, 0):   # captured on frame 0; later replays see other cameras / expressions, then come back
        out = graphed(ws, cams[i:i + 1], uvs[i:i + 1], jits[i:i + 1])
        torch.cuda.synchronize()
        assert torch.equal(out['image'], want[i][0]), i
        assert torch.equal(out['image_depth'], want[i][1]), i
    assert not torch.equal(want[0][0], want[2][0])   # (the frames do differ)


def test_frames_in_flight_return_their_own_results():
    frames = [3, 60, 150, 230, 90]
    g, ws, cams, uvs, jits, eager = _setup('small', 64, frames)
    want = [eager(i)[0] for i in range(len(frames))]
    pipe = FramePipeline(g, depth=2, batch=1, neural_rendering_resolution=64).capture(ws, cams[:1], uvs[:1], jits[:1])
    got = []
    for i in range(len(frames)):
        out, ev, _ = pipe.submit(ws, cams[i:i + 1], uvs[i:i + 1], jits[i:i + 1])
        ev.synchronize()              # the slot's output buffer is rewritten two submits later: take it now
        got.append(out['image'].clone())
    pipe.drain()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), i


def test_bench_workload_replay_equals_eager():
    """BASELINE configs[1]: full-width model, 128^2 neural render, 512^2 out, B = 1 -- the frame bench.py times."""
    frames = [0, 16]
    g, ws, cams, uvs, jits, eager = _setup('full', 128, frames)
    want = [eager(i) for i in range(2)]
    graphed = GraphedSynthesis(g, batch=1, neural_rendering_resolution=128)
    for i in (0, 1, 0):
        out = graphed(ws, cams[i:i + 1], uvs[i:i + 1], jits[i:i + 1])
        torch.cuda.synchronize()
        assert out['image'].shape == (1, 3, 512, 512)
        assert torch.equal(out['image'], want[i][0]), i
        assert torch.equal(out['image_depth'], want[i][1]), i
    assert torch.isfinite(want[0][0]).all() and want[0][0].abs().mean().item() > 1e-3


def test_back_to_back_frames_do_not_interfere():
    """Eager frames and graph replays issued without a host sync in between (the side streams of frame k+1 are forked while
    frame k is still running) give the bits of frames run one at a time."""
    frames = list(range(0, 240, 20))
    g, ws, cams, uvs, jits, eager = _setup('small', 64, frames)
    want = [eager(i)[0] for i in range(len(frames))]
    torch.cuda.synchronize()
    got = []
    with torch.no_grad():
        for k in range(36):
            i = (k * 5) % len(frames)
            out = g.synthesis(ws, cams[i:i + 1], {'uvcoords_image': uvs[i:i + 1]}, neural_rendering_resolution=64, noise_mode='const',
                              evaluation=True, jitter=jits[i:i + 1])
            got.append((i, out['image']))
    graphed = GraphedSynthesis(g, batch=1, neural_rendering_resolution=64)
    for k in range(60):
        i = (k * 7) % len(frames)
        got.append((i, graphed(ws, cams[i:i + 1], uvs[i:i + 1], jits[i:i + 1])['image'].clone()))
    torch.cuda.synchronize()
    assert sum(int(not torch.equal(a, want[i])) for i, a in got) == 0


def test_generator_copies_and_pickles_after_a_device_forward_pass():
    """streams / per-frame tensors must not live in the module: deepcopy, pickle and torch.save of a generator that
    has rendered on the device work and the copies render the same bits."""
    import copy, io, pickle
    g = TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)
    synthetic.fill_parameters(g)
    g = g.cuda()
    frames, nrr = [3], 32
    with torch.no_grad():
        ws = g.mapping(synthetic.latent(0, 1).cuda(), synthetic.conditioning_camera().cuda(), truncation_psi=0.7, truncation_cutoff=14)
        args = (ws, synthetic.camera_labels(frames).cuda(), {'uvcoords_image': synthetic.uv_conditions(frames).cuda()})
        kw = dict(neural_rendering_resolution=nrr, noise_mode='const', evaluation=True, jitter=synthetic.jitter(frames, nrr * nrr).cuda())
        ref = g.synthesis(*args, **kw)['image']
        buf = io.BytesIO()
        torch.save(g, buf)
        buf.seek(0)
        for other in (copy.deepcopy(g), pickle.loads(pickle.dumps(g)), torch.load(buf, weights_only=False)):
            assert torch.equal(other.synthesis(*args, **kw)['image'], ref)


def test_config4_shard_of_eight_frames_full_width():
    """BASELINE configs[3] as one rank sees it: 8 full-width frames in ONE synthesis call (batch-global `dist` passed in) against
    the same 8 frames rendered one per call with that `dist`; and the captured graph of the 8-frame call against the eager call
    (bit for bit).  Between batch sizes the stream-K split of the convolutions differs (ia_conv2d_plan depends on B), so the
    frames agree to fp32 summation-order level, not bitwise."""
    from invertavatar_amd import frame_parallel
    nrr, frames = 128, list(range(16, 24))
    g, ws, cams, uvs, jits, _ = _setup('full', nrr, frames)
    dist = frame_parallel.global_ray_dist(synthetic.camera_labels(list(range(64))).cuda())     # of a B = 64 batch
    with torch.no_grad():
        call = lambda sl: g.synthesis(ws.expand(cams[sl].shape[0], -1, -1), cams[sl], {'uvcoords_image': uvs[sl]}, neural_rendering_resolution=nrr, noise_mode='const',   # noqa: E731
                                      evaluation=True, jitter=jits[sl], ray_dist=dist)['image']
        batched = call(slice(0, 8)).clone()
        for i in range(8):
            single = call(slice(i, i + 1))
            assert (batched[i:i + 1] - single).abs().max().item() <= 2e-5, i
        graphed = GraphedSynthesis(g, batch=8, neural_rendering_resolution=nrr, with_ray_dist=True)
        replay = graphed(ws, cams, uvs, jits, dist)['image'].clone()
        assert torch.equal(replay, batched)
        replay2 = graphed(ws, cams.flip(0), uvs.flip(0), jits.flip(0), dist)['image']
        assert torch.equal(replay2, call(slice(0, 8)).flip(0)) or (replay2 - batched.flip(0)).abs().max().item() <= 2e-5
