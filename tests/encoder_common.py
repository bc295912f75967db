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


def run_few_shot(net, device, nrr=32):
    """encode -> two AR_eval_forward groups with carried GRU state -> one drive frame, through the product's eval_seq harness
    (sequential groups: the fixture's two groups are given in the order they are consumed)."""
    from invertavatar_amd import eval_seq
    net.generator.neural_rendering_resolution = nrr
    groups, drive = encoder_inputs(nrr)
    cat = lambda key: torch.cat([g[key] for g in groups]).to(device)
    ws, res, r_list = eval_seq.few_shot_inversion(net, cat('image'), cat('uv'), cat('c'), cat('uvcoords'), sequential_sampling=True, chain_results=True,
                                                  hook=lambda idx: fixed_randomness(groups[idx]['jitter']))
    with fixed_randomness(drive['jitter']):
        image, _ = eval_seq.drive_sequence(net, ws, res, drive['c'].to(device), drive['uvcoords'].to(device))
    return ws, res, r_list, image


def compare_with_fixture(gld, ws, res, r_list, image, tol):
    import re
    worst = {}
    def check(prefix, t):
        key = [k for k in gld.keys() if re.fullmatch(re.escape(prefix) + r'_s\d+', k)][0]
        s = int(key.rsplit('_s', 1)[1])
        ref = gld[key]
        got = t.detach().float().cpu()[..., ::s, ::s]
        assert got.shape == ref.shape, (prefix, got.shape, ref.shape)
        scale = max(ref.abs().max().item(), 1.0)
        worst[prefix] = (got - ref).abs().max().item() / scale
    for i, t in enumerate(res['texture']): check(f'texture{i}', t)
    for i, t in enumerate(res['static']): check(f'static{i}', t)
    for u, states in enumerate(r_list):
        for k, h in enumerate(states): check(f'gru{u}_{k}', h)
    check('drive_image', image)
    ws_err = (ws.cpu() - gld['ws']).abs().max().item()
    bad = {k: v for k, v in worst.items() if v > tol}
    assert ws_err <= tol and not bad, (ws_err, bad)
    return max(worst.values())
