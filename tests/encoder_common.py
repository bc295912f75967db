"""Shared driver for the few-shot inversion tests: same inputs / module modes / noise pinning as tests/golden/make_golden.py."""
import contextlib

import numpy as np
import torch

from invertavatar_amd import synthetic


@contextlib.contextmanager
def fixed_randomness(jit, u_seed=99):
    orig_like, orig_rand = torch.rand_like, torch.rand

    def fake_like(t, *a, **k):
        assert tuple(t.shape) == tuple(jit.shape), (t.shape, jit.shape)
        return jit.to(device=t.device, dtype=t.dtype)

    def fake_rand(*size, **k):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        t = torch.from_numpy(np.random.RandomState(u_seed).rand(*shape).astype(np.float32))
        return t.to(k['device']) if k.get('device') is not None else t
    torch.rand_like, torch.rand = fake_like, fake_rand
    try:
        yield
    finally:
        torch.rand_like, torch.rand = orig_like, orig_rand


def encoder_inputs(nrr=32):
    groups = [[0, 8, 16, 24], [4, 12, 20, 28]]
    data = [dict(image=synthetic.source_frames(7 + gi, 4), uv=synthetic.source_uv(17 + gi, fr), c=synthetic.camera_labels(fr),
                 uvcoords=synthetic.uv_conditions(fr), jitter=synthetic.jitter(fr, nrr * nrr)) for gi, fr in enumerate(groups)]
    drive = [40]
    return data, dict(c=synthetic.camera_labels(drive), uvcoords=synthetic.uv_conditions(drive), jitter=synthetic.jitter(drive, nrr * nrr))


from invertavatar_amd.eval_seq import set_eval_seq_modes  # noqa: E402,F401  (module modes of eval_seq.py:92-97)


def build_inversion_net(width='full'):
    from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
    from invertavatar_amd.encoder_inversion.models.uvnet import inversionNet
    g = TriPlaneGenerator(**synthetic.generator_kwargs(width)).eval().requires_grad_(False)
    synthetic.fill_parameters(g)
    net = inversionNet(generator=g, encoding_triplane=True, encoding_texture=True).requires_grad_(False)
    synthetic.fill_encoder_parameters(net)
    return set_eval_seq_modes(net)


def drive_frame2(nrr=128):
    fr = [47]
    return dict(c=synthetic.camera_labels(fr), uvcoords=synthetic.uv_conditions(fr), jitter=synthetic.jitter(fr, nrr * nrr))


def source_batch(device):
    """The 8 sources of the fixture in the order the script holds them (two recorded clips of four frames each)."""
    groups, _ = encoder_inputs()
    return {k: torch.cat([g[k] for g in groups]).to(device) for k in groups[0]}


def run_few_shot(net, device, nrr=32, nrr_drive2=128):
    """The DEFAULT path of the product's eval_seq harness, as the reference script runs it on 8 sources (eval_seq.py:168-212):
    encode -> two interleaved AR_eval_forward groups ([idx::2]), each started from the e4e features, ConvGRU state carried ->
    drive frames from the last group's features (one at the inversion's nrr, one at the deployed nrr 128)."""
    from invertavatar_amd import eval_seq
    net.generator.neural_rendering_resolution = nrr
    _, drive = encoder_inputs(nrr)
    src = source_batch(device)
    n_it = src['image'].shape[0] // 4
    ws, res, r_list = eval_seq.few_shot_inversion(net, src['image'], src['uv'], src['c'], src['uvcoords'],
                                                  hook=lambda idx: fixed_randomness(src['jitter'][idx::n_it]))
    with fixed_randomness(drive['jitter']):
        image, _ = eval_seq.drive_sequence(net, ws, res, drive['c'].to(device), drive['uvcoords'].to(device))
    image2 = None
    if nrr_drive2:
        d2 = drive_frame2(nrr_drive2)
        with fixed_randomness(d2['jitter']):
            image2, _ = eval_seq.drive_sequence(net, ws, res, d2['c'].to(device), d2['uvcoords'].to(device), neural_rendering_resolution=nrr_drive2)
        net.generator.neural_rendering_resolution = nrr
    return ws, res, r_list, image, image2


def fixture_deviations(gld, ws, res, r_list, image, image2=None):
    """{name: max |got - ref| / max(1, max |ref|)} for every recorded tensor of encoder_fewshot.npz."""
    import re
    worst = {}

    def check(prefix, t):
        key = [k for k in gld.keys() if re.fullmatch(re.escape(prefix) + r'_s\d+', k)][0]
        s = int(key.rsplit('_s', 1)[1])
        ref = gld[key]
        got = t.detach().float().cpu()[..., ::s, ::s]
        assert got.shape == ref.shape, (prefix, got.shape, ref.shape)
        worst[prefix] = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1.0)
    for i, t in enumerate(res['texture']): check(f'texture{i}', t)
    for i, t in enumerate(res['static']): check(f'static{i}', t)
    for u, states in enumerate(r_list):
        for k, h in enumerate(states): check(f'gru{u}_{k}', h)
    check('drive_image', image)
    if image2 is not None:
        check('drive_image_nrr128', image2)
        img = image2.detach().float().cpu()
        worst['drive_image_nrr128_crop'] = (img[:, :, 192:320, 192:320] - gld['drive_image_nrr128_crop']).abs().max().item()
        worst['drive_image_nrr128_blockmean'] = (torch.nn.functional.avg_pool2d(img.double(), 32).float()
                                                 - gld['drive_image_nrr128_blockmean']).abs().max().item()
    worst['ws'] = (ws.cpu() - gld['ws']).abs().max().item()
    return worst


def compare_with_fixture(gld, ws, res, r_list, image, tol, image2=None):
    worst = fixture_deviations(gld, ws, res, r_list, image, image2)
    bad = {k: v for k, v in worst.items() if v > tol}
    assert not bad, bad
    return max(worst.values())
