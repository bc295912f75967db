"""Improved one-shot inversion encoders of eval_updated_os.py (uvnet_new.inversionNet with the SegFormer-style
UNet decoders): checkpoint-name compatibility and the eval_updated_os flow against the reference fixture
(tests/golden/make_golden.py:gen_encoder_new)."""
import os

import pytest
import torch

from invertavatar_amd import eval_updated_os, synthetic
from conftest import GOLDEN
from encoder_common import compare_with_fixture, fixed_randomness  # noqa: F401


TOL_ONESHOT = 5e-5      # relative to max(1, max |ref|); the flow now runs ia_attention, the HIP trunk / decoder convolutions (measured r03: see the test's print)


def _build(device):
    from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
    from invertavatar_amd.encoder_inversion.models.uvnet_new import inversionNet
    g = TriPlaneGenerator(**synthetic.generator_kwargs('full')).eval().requires_grad_(False)
    synthetic.fill_parameters(g)
    net = inversionNet(generator=g, encoding_triplane=True, encoding_texture=True).eval().requires_grad_(False)
    synthetic.fill_encoder_parameters(net)
    return net.to(device)


def test_state_dict_names_match_the_reference():
    from invertavatar_amd.encoder_inversion.models.uvnet_new import improved_os_unet_encoder
    enc = improved_os_unet_encoder(encoding_texture=True, encoding_triplane=True)
    mine = {f'unet_encoder.{n}': (tuple(t.shape), str(t.dtype).replace('torch.', '')) for n, t in enc.state_dict().items()}
    ref = {}
    for line in open(os.path.join(GOLDEN, 'encoder_new_state_names.txt')):
        n, s, d = line.rstrip('\n').split('\t')
        ref[n] = (eval(s), d)
    assert set(mine) == set(ref), (sorted(set(ref) - set(mine))[:5], sorted(set(mine) - set(ref))[:5])
    assert mine == ref


def _run(net, device, nrr=32):
    src, drive = [12], [40]
    to = lambda t: t.to(device)   # noqa: E731
    net.generator.neural_rendering_resolution = nrr
    with fixed_randomness(synthetic.jitter(src, nrr * nrr)):
        ws, res = eval_updated_os.one_shot_inversion(net, to(synthetic.source_frames(9, 1)), to(synthetic.source_uv(19, src)),
                                                     to(synthetic.camera_labels(src)), to(synthetic.uv_conditions(src)))
    with fixed_randomness(synthetic.jitter(drive, nrr * nrr)), torch.no_grad():
        img = net.generator.synthesis_withTexture(ws, res['texture'], to(synthetic.camera_labels(drive)),
                                                  {'uvcoords_image': to(synthetic.uv_conditions(drive))}, noise_mode='const',
                                                  static_feats=res['static'], evaluation=True)['image']
    return ws, res, img


def _compare(gld, ws, res, img, tol):
    import re
    worst = {}

    def check(prefix, t):
        key = [k for k in gld.keys() if re.fullmatch(re.escape(prefix) + r'_s\d+', k)][0]
        s = int(key.rsplit('_s', 1)[1])
        ref = gld[key]
        got = t.detach().float().cpu()[..., ::s, ::s]
        assert got.shape == ref.shape, (prefix, got.shape, ref.shape)
        worst[prefix] = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1.0)
    for i, t in enumerate(res['texture']):
        check(f'texture{i}', t)
    check('static5', res['static'][-1])
    check('drive_image', img)
    ws_err = (ws.cpu() - gld['ws']).abs().max().item()
    bad = {k: v for k, v in worst.items() if v > tol}
    assert ws_err <= tol and not bad, (ws_err, bad)
    return max(worst.values())


@pytest.mark.gpu
def test_one_shot_inversion_matches_reference(golden):
    net = _build('cuda')
    worst = _compare(golden('encoder_oneshot.npz'), *_run(net, 'cuda'), tol=TOL_ONESHOT)
    print(f'one-shot inversion (eval_updated_os flow): worst relative deviation {worst:.2e}')


@pytest.mark.gpu
@pytest.mark.parametrize('b,h,w,c', [(1, 64, 64, 2048), (2, 8, 8, 2048), (1, 5, 7, 12)])
def test_depthwise_token_convolution_matches_the_module(b, h, w, c):
    """ia_dwconv3x3_tokens (Mix-FFN's depth-wise 3x3 on the channels-last token grid, optionally with the GELU behind it) against
    mix_transformer.DWConv's own torch route in fp64 (mix_transformer.py:49-58, :70-77)."""
    import copy
    from invertavatar_amd.encoder_inversion.models.mmseg import mix_transformer
    torch.manual_seed(c + h)
    mod = mix_transformer.DWConv(c).requires_grad_(False)
    mod.dwconv.weight.normal_(0, 0.4); mod.dwconv.bias.normal_(0, 0.2)
    x = torch.randn(b, h * w, c)
    ref = copy.deepcopy(mod).double()
    want = ref(x.double(), h, w)
    want_gelu = torch.nn.functional.gelu(want)
    mod = mod.cuda()
    with torch.no_grad():
        got, got_gelu = mod(x.cuda(), h, w).cpu(), mod(x.cuda(), h, w, gelu=True).cpu()
        mix_transformer.HIP_DWCONV = False
        try:
            lib = mod(x.cuda(), h, w, gelu=True).cpu()
        finally:
            mix_transformer.HIP_DWCONV = True
    for name, a, ref_t in (('linear', got, want), ('gelu', got_gelu, want_gelu), ('library', lib, want_gelu)):
        err = (a.double() - ref_t).abs().max().item() / max(ref_t.abs().max().item(), 1.0)
        print(f'dwconv tokens B{b} {h}x{w} C{c} [{name}]: {err:.2e}')
        assert a.shape == (b, h * w, c) and err <= 2e-6


@pytest.mark.gpu
def test_graphed_one_shot_inversion_equals_the_eager_flow():
    """eval_updated_os.GraphedOneShot (the whole one-shot inversion as one hipGraph, bench.py's oneshot leg) against the eager flow on
    the source it was captured with and on another one; the renderer's random draws pinned to device tensors made before the capture."""
    import contextlib
    import numpy as np
    net = _build('cuda')
    nrr = 32
    net.generator.neural_rendering_resolution = nrr
    jit = synthetic.jitter([12], nrr * nrr).cuda()
    draws = {}

    @contextlib.contextmanager
    def device_randomness():
        orig_like, orig_rand = torch.rand_like, torch.rand

        def fake_like(t, *a, **k):
            assert tuple(t.shape) == tuple(jit.shape)
            return jit.to(t.dtype)

        def fake_rand(*size, **k):
            shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
            if shape not in draws:
                draws[shape] = torch.from_numpy(np.random.RandomState(99).rand(*shape).astype(np.float32)).cuda()
            return draws[shape]
        torch.rand_like, torch.rand = fake_like, fake_rand
        try:
            yield
        finally:
            torch.rand_like, torch.rand = orig_like, orig_rand

    def inputs(seed, frame):
        return (synthetic.source_frames(seed, 1).cuda(), synthetic.source_uv(seed + 10, [frame]).cuda(), synthetic.camera_labels([frame]).cuda(),
                synthetic.uv_conditions([frame]).cuda())
    first, second = inputs(9, 12), inputs(4, 27)
    with device_randomness():
        replay = eval_updated_os.GraphedOneShot(net, *first)
        for inp in (first, second, first):
            ws_e, res_e = eval_updated_os.one_shot_inversion(net, *inp)
            ws_g, res_g = replay(*inp)
            worst = (ws_g - ws_e).abs().max().item()
            for a, b in zip(res_g['texture'] + res_g['static'], res_e['texture'] + res_e['static']):
                worst = max(worst, (a - b).abs().max().item() / max(b.abs().max().item(), 1.0))
            print(f'graph replay vs eager: worst relative deviation {worst:.2e}')
            assert worst <= 5e-6          # (library GEMMs may pick other kernels under capture; the fixture tolerance is TOL_ONESHOT)


@pytest.mark.gpu
@pytest.mark.parametrize('b,n,m', [(1, 64, 64), (2, 256, 256), (1, 512, 256), (1, 96, 40)])
def test_fused_attention_matches_the_torch_definition(b, n, m):
    """ia_attention (softmax(QK^T * scale) V per head, online softmax, no score matrix) against mix_transformer.Attention's own
    torch arithmetic in fp64 (mix_transformer.py:83-116); head_dim 256, 4 heads: the transformer_block configuration.  (96, 40):
    token counts that are not multiples of the 32-row tiles, keys != queries."""
    from invertavatar_amd import hipops
    torch.manual_seed(n + m)
    heads, hd = 4, 256
    c = heads * hd
    q = torch.randn(b, n, c) * 0.7
    kv = torch.randn(b, m, 2 * c) * 0.7
    scale = hd ** -0.5
    qh = q.double().reshape(b, n, heads, hd).permute(0, 2, 1, 3)
    k, v = kv.double().reshape(b, m, 2, heads, hd).permute(2, 0, 3, 1, 4)
    want = (((qh @ k.transpose(-2, -1)) * scale).softmax(dim=-1) @ v).transpose(1, 2).reshape(b, n, c).float()
    got = hipops.attention(q.cuda(), kv.cuda(), heads, scale).cpu()
    again = hipops.attention(q.cuda(), kv.cuda(), heads, scale).cpu()
    assert torch.equal(got, again)
    err = (got - want).abs().max().item()
    print(f'attention B{b} N{n} M{m}: max |d| = {err:.2e}')
    assert err <= 5e-6


@pytest.mark.gpu
def test_attention_module_routes_through_the_kernel():
    from invertavatar_amd.encoder_inversion.models.mmseg import mix_transformer as MT
    torch.manual_seed(0)
    att = MT.Attention(1024, num_heads=4, qkv_bias=True).eval().requires_grad_(False)
    x = torch.randn(1, 256, 1024)
    want = att.double()(x.double(), 16, 16).float()
    att = att.float().cuda()
    with torch.no_grad():
        got = att(x.cuda(), 16, 16).cpu()
        MT.HIP_ATTENTION = False
        try:
            lib = att(x.cuda(), 16, 16).cpu()
        finally:
            MT.HIP_ATTENTION = True
    assert (got - want).abs().max().item() <= 2e-5 and (lib - want).abs().max().item() <= 2e-5
    assert not torch.equal(got, lib)            # (two different code paths ran)
