"""Encoder front end on CPU: checkpoint-compatible names and the stand-alone stages against the
reference fixture.  The full few-shot flow runs in the GPU suite (tests/test_encoder_gpu.py)."""
import os

import torch

from invertavatar_amd import synthetic
from conftest import GOLDEN, max_abs


def test_encoder_state_names_match_reference():
    from invertavatar_amd.encoder_inversion.models.uvnet import inversionNet
    from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
    g = TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)
    net = inversionNet(generator=g, encoding_triplane=True, encoding_texture=True)
    mine = {n: (tuple(t.shape), str(t.dtype).replace('torch.', '')) for n, t in net.state_dict().items() if not n.startswith('generator.')}
    ref = {}
    for line in open(os.path.join(GOLDEN, 'encoder_state_names.txt')):
        n, s, d = line.rstrip('\n').split('\t')
        ref[n] = (eval(s), d)
    assert len(ref) == 1519 and mine == ref
    assert len(net.state_dict()) == 1519 + 444          # 1 963 tensors with the generator


def test_e4e_encode_matches_reference(golden):
    """encode() = face pool + IR-SE50 + 14 style heads + latent average (uvnet.py:107-115)."""
    from encoder_common import build_inversion_net, encoder_inputs
    net = build_inversion_net('small')     # encode() does not touch the generator beyond w_avg... which differs by width:
    from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
    # the fixture was produced with the full-width generator; w_avg is filled by name, identical in both widths
    groups, _ = encoder_inputs()
    with torch.no_grad():
        ws = net.encode(groups[0]['image'][:1])
    assert max_abs(ws, golden('encoder_fewshot.npz')['ws']) <= 2e-4


def test_convgru_and_unet_shapes():
    from invertavatar_amd.encoder_inversion.models.unet_encoders import ConvGRU, TriPlanefeat_Encoder
    gru = ConvGRU(8)
    x = torch.randn(2, 3, 8, 5, 5)
    o, h = gru(x, None)
    assert o.shape == (2, 8, 5, 5) and torch.equal(o, h)
    o2, h2 = gru(x, h, seq2seq=True)
    assert o2.shape == (2, 3, 8, 5, 5) and torch.equal(o2[:, -1], h2)
    # GRU update is a convex combination of the state and a tanh candidate -> bounded by 1 when started from zero
    assert h2.abs().max() <= 1.0
