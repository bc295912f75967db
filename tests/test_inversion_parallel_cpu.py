"""inversion_parallel.few_shot_inversion_sharded over two gloo ranks on CPU: the orchestration --
frame-sharded source renders with each group's depth range and random draws, one all-gather, the two UNet chains on ranks 0 / 1
with their ConvGRU states carried from group to group, one broadcast per owner -- against the one-process flow.

The real inversion network needs the full-width generator (minutes per run on CPU: its device test is tests/test_encoder_gpu.py);
here a miniature stand-in with the same interfaces runs through the REAL `inversionNet.AR_eval_forward` / `get_unet_uvinput` code:
every stage depends on all of its inputs (cameras, draws, depth range, all four frames of a group, the recurrent state), so a
mis-routed frame, draw or state changes the result."""
import contextlib
import os

import pytest
import torch
import torch.multiprocessing as mp

from invertavatar_amd import eval_seq, frame_parallel, inversion_parallel, synthetic
from invertavatar_amd.encoder_inversion.models.uvnet import inversionNet

RES, NRR = 32, 4


class _Backbone(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.maps = [torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 6, 16, 16, generator=g)]

    def synthesis(self, ws, cond_list=None, return_list=True, feat_conditions=None, update_emas=False, noise_mode='const'):
        out = [m * ws.mean() for m in self.maps]
        if feat_conditions is not None:
            out = [o * (1 + c) for o, c in zip(out, feat_conditions)]
        return out


class _Generator(torch.nn.Module):
    img_resolution, neural_rendering_resolution = RES, NRR

    def __init__(self):
        super().__init__()
        self.texture_backbone, self.backbone = _Backbone(1), _Backbone(2)

    def synthesis_withTexture(self, ws, texture_feats, c, mesh_condition, static_feats=None, noise_mode='const',
                              neural_rendering_resolution=None, jitter=None, u_importance=None, ray_dist=None, evaluation=False):
        b, r = c.shape[0], (neural_rendering_resolution or NRR) ** 2
        if jitter is None:          # the renderer's own draws, in its call shapes and order (renderer.py:406, :453)
            jitter = torch.rand_like(torch.empty(b, r, 48, 1))
            u_importance = torch.rand(b * r, 48)
        if ray_dist is None:
            ray_dist = frame_parallel.global_ray_dist(c)
        stat = (jitter.reshape(b, -1).mean(1) + 3 * u_importance.reshape(b, -1).sort(dim=-1).values[:, ::7].mean(1)).reshape(b, 1, 1, 1)
        base = sum(f.mean() for f in texture_feats) + sum(f.mean() for f in static_feats)
        cam = c[:, :16].sum(1).reshape(b, 1, 1, 1)
        uvm = mesh_condition['uvcoords_image'].mean(dim=(1, 2, 3)).reshape(b, 1, 1, 1)
        ramp = torch.linspace(0, 1, RES * RES * 3).reshape(1, 3, RES, RES)
        return {'image': torch.tanh(ramp * (stat + cam * 0.1 + uvm) + base + ray_dist.reshape(-1)[0] * 0.05)}


class _UNet(torch.nn.Module):
    """Stand-in for a ConvGRU UNet.  Like the real ones it is a per-frame trunk (`forward_onlyEncoder`: nothing couples the frames,
    the IR-SE50 trunks run eval-mode BatchNorm) followed by a decoder whose per-level outputs mix ALL frames of the group (like its
    train-mode BatchNorm) with a recurrent state that the next group continues from (`forward_onlyDecoder`)."""

    def __init__(self, in_ch, levels, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w = [torch.randn(ch, in_ch, generator=g) * 0.2 for ch, _ in levels]
        self.levels = levels
        # the attributes the product inspects for the mode of the trunk BatchNorms (unet_encoders.py:162); identity here
        self.input_layer, self.body = torch.nn.Sequential(torch.nn.BatchNorm2d(in_ch)).eval(), torch.nn.Sequential().eval()

    def forward_onlyEncoder(self, x):
        t = x.flatten(0, 1)                                 # [T, C, H, W]
        return [torch.einsum('oc,tchw->tohw', w, torch.tanh(torch.nn.functional.adaptive_avg_pool2d(t, res))) for w, (ch, res) in zip(self.w, self.levels)]

    def gru_state_shapes(self, feats):                      # (what the product's UNets answer from their trunk features)
        return [(1, ch, res, res) for ch, res in self.levels]

    def forward_onlyDecoder(self, T, feats, r_list=None):
        if r_list is None:
            r_list = [torch.zeros(1, ch, res, res) for ch, res in self.levels]
        outs, states = [], []
        for f, h in zip(feats, r_list):
            assert f.shape[0] == T
            mixed = f * torch.arange(1, T + 1).reshape(-1, 1, 1, 1)
            new_h = torch.tanh(0.5 * h + mixed.mean(0, keepdim=True) - mixed.std(0, keepdim=True))
            outs.append(new_h * 0.3)
            states.append(new_h)
        return outs, states

    def forward(self, x, r_list=None, return_list=True):
        return self.forward_onlyDecoder(x.shape[1], self.forward_onlyEncoder(x), r_list)


class _Toy(torch.nn.Module):
    AR_eval_forward = inversionNet.AR_eval_forward          # the product's own group update and UV-space residual
    get_unet_uvinput = inversionNet.get_unet_uvinput
    trunk_features = inversionNet.trunk_features           # ... and the per-frame half the sharded flow deals to the ranks
    trunks_in_eval_mode = inversionNet.trunks_in_eval_mode
    require_eval_trunks = inversionNet.require_eval_trunks

    def __init__(self):
        super().__init__()
        self.generator = _Generator()
        self.unet_encoder = torch.nn.Module()
        self.unet_encoder.texture_unet = _UNet(7, [(4, 8), (6, 16)], 3)
        self.unet_encoder.triplane_unet = _UNet(6, [(4, 8), (6, 16)], 4)
        self.black_uv_bg = torch.zeros(1, 3, 1, 1) - 1

    def encode(self, x):
        return x.mean().reshape(1, 1, 1).expand(1, 14, 8) + torch.linspace(0.5, 1.5, 14 * 8).reshape(1, 14, 8)


def _inputs(s=8):
    g = torch.Generator().manual_seed(5)
    frames = list(range(0, 4 * s, 4))
    cams = synthetic.camera_labels(frames)
    for k in range(s):
        cams[k, [3, 7, 11]] *= 1.0 + 0.04 * k                 # groups get different depth ranges
    images = torch.rand(s, 3, RES, RES, generator=g) * 2 - 1
    uvs = torch.cat([torch.rand(s, 3, RES, RES, generator=g) * 2 - 1, torch.rand(s, 2, RES, RES, generator=g) * 2 - 1,
                     (torch.rand(s, 1, RES, RES, generator=g) > 0.3).float()], 1)
    uvcoords = torch.rand(s, 16, 16, 3, generator=g)
    return images, uvs, cams, uvcoords


def _worker(rank, world, port, tmp, shard_trunks):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    out = inversion_parallel.few_shot_inversion_sharded(_Toy(), *_inputs(), rank=rank, world_size=world,
                                                       draws=inversion_parallel.seeded_draws(7, NRR * NRR), shard_trunks=shard_trunks)
    torch.save(out, f'{tmp}.{rank}')
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _same(a, b, tol=0.0):
    ws_a, res_a, r_a = a
    ws_b, res_b, r_b = b
    pairs = [(ws_a, ws_b)] + list(zip(res_a['texture'], res_b['texture'])) + list(zip(res_a['static'], res_b['static']))
    pairs += [(x, y) for sa, sb in zip(r_a, r_b) for x, y in zip(sa, sb)]
    assert len(res_a['texture']) == len(res_b['texture']) == 2 and len(r_a[0]) == len(r_b[0]) == 2
    return max((x - y).abs().max().item() for x, y in pairs) <= tol, [(x - y).abs().max().item() for x, y in pairs]


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world,shard_trunks', [(2, True), (3, True), (2, False), (5, True)])
def test_sharded_inversion_reproduces_the_one_process_flow(tmp_path, world, shard_trunks):
    """shard_trunks: the UNet trunks dealt to the ranks by frame with the renders (r05) or run whole on the chains' owners (r04).
    5 ranks on 8 frames: uneven blocks, and the frames of one group spread over three ranks."""
    tmp = str(tmp_path / 'out')
    mp.spawn(_worker, args=(world, 29651 + world + (10 if not shard_trunks else 0), tmp, shard_trunks), nprocs=world, join=True)
    draws = inversion_parallel.seeded_draws(7, NRR * NRR)
    one = inversion_parallel.few_shot_inversion_sharded(_Toy(), *_inputs(), rank=0, world_size=1, draws=draws)
    for rank in range(world):                              # every rank ends with the same features and states
        ok, devs = _same(torch.load(f'{tmp}.{rank}'), one, tol=1e-6)
        assert ok, (rank, devs)

    # ... and the one-rank form is eval_seq.few_shot_inversion's flow (groups interleaved, every group from the e4e features,
    # ConvGRU states carried) when the renderer draws what `draws` hands out
    @contextlib.contextmanager
    def pinned(idx):
        jit, u = draws(idx)
        orig_like, orig_rand = torch.rand_like, torch.rand
        torch.rand_like = lambda t, *a, **k: jit.reshape(t.shape)
        torch.rand = lambda *size, **k: u.reshape(size)
        try:
            yield
        finally:
            torch.rand_like, torch.rand = orig_like, orig_rand
    script = eval_seq.few_shot_inversion(_Toy(), *_inputs(), hook=pinned)
    ok, devs = _same(script, one, tol=1e-5)
    assert ok, devs
    # a mis-routed draw would show: other draws give other features
    other = inversion_parallel.few_shot_inversion_sharded(_Toy(), *_inputs(), draws=inversion_parallel.seeded_draws(8, NRR * NRR))
    assert not _same(other, one, tol=1e-4)[0]


def test_train_mode_trunks_are_not_dealt_by_frame():
    """per-frame trunk passes equal the one-process result only with eval-mode trunk BatchNorms (eval_seq.py:96-97).
    trunk_features refuses a train-mode trunk, and the sharded flow then keeps the groups whole on the chains' owners."""
    net = _Toy()
    assert net.trunks_in_eval_mode()
    net.unet_encoder.texture_unet.input_layer.train()
    assert not net.trunks_in_eval_mode()
    images, uvs, cams, uvcoords = _inputs(4)
    with pytest.raises(RuntimeError, match='eval mode'):
        net.trunk_features(images[:1], uvs[:1], images[:1])
