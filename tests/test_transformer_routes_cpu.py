"""CPU checks of the host side of the transformer-block routes (r06): the restructured Block / Attention / Mlp / OverlapPatchEmbed forward
passes (LayerNorm and residual handed to the sub-modules so that the device path can fuse them) compute what the reference's composition
computes (encoder_inversion/models/mmseg/mix_transformer.py:83-190), and the weight packings of the fp16-pair GEMM reconstruct the weights."""
import pytest
import torch

from invertavatar_amd import hipops
from invertavatar_amd.encoder_inversion.models.mmseg import mix_transformer as mt


@pytest.mark.parametrize('sr_ratio', [1, 2])
def test_block_forward_equals_the_reference_composition(sr_ratio):
    torch.manual_seed(0)
    blk = mt.Block(dim=64, num_heads=4, mlp_ratio=2, sr_ratio=sr_ratio).eval()
    x = torch.randn(2, 16, 64)
    with torch.no_grad():
        got = blk(x, 4, 4)
        ref = x + blk.attn(blk.norm1(x), 4, 4)                      # Block.forward of the reference: x + attn(norm1(x)); x + mlp(norm2(x))
        ref = ref + blk.mlp(blk.norm2(ref), 4, 4)
        assert torch.equal(got, ref)
        # the keyword forms the device path uses
        a = blk.attn(x, 4, 4, residual=x, norm=blk.norm1)
        assert torch.allclose(a, x + blk.attn(blk.norm1(x), 4, 4), atol=1e-6)
        m = blk.mlp(a, 4, 4, residual=a, norm=blk.norm2)
        assert torch.allclose(m, got, atol=1e-6)


def test_patch_embed_and_mlp_head_on_cpu():
    torch.manual_seed(1)
    pe = mt.OverlapPatchEmbed(img_size=0, stride=2, in_chans=8, embed_dim=16).eval()
    x = torch.randn(1, 8, 12, 10)
    with torch.no_grad():
        tokens, h, w = pe(x)
        ref = pe.norm(pe.proj(x).flatten(2).transpose(1, 2))
    assert (h, w) == (6, 5) and torch.equal(tokens, ref)
    head = mt.MLP(input_dim=8, embed_dim=4).eval()
    with torch.no_grad():
        assert torch.equal(head(x), head.proj(x.flatten(2).transpose(1, 2)))


def test_linear_weight_split_reconstructs_the_weight():
    torch.manual_seed(2)
    w = torch.randn(24, 48) * 0.05
    ws = hipops.pack_linear_weight_split(w)
    assert ws.shape == (2, 1, 6, 24, 8) and ws.dtype == torch.float16
    back = (ws[0].float() + ws[1].float())[0].permute(1, 0, 2).reshape(24, 48) * 2.0 ** -ws.wk_exp      # [K/8][N][8] -> [N][K]
    assert (back - w).abs().max().item() <= 2.0 ** -21 * w.abs().max().item()
    # a convolution weight as a patch matrix: columns (c, ky, kx), zero-padded to a multiple of 16
    cw = torch.randn(10, 3, 7, 7)
    pw = hipops.pack_patch_weight_split(cw)
    assert pw.shape == (2, 1, 20, 10, 8)                            # 147 -> 160 columns
    flat = (pw[0].float() + pw[1].float())[0].permute(1, 0, 2).reshape(10, 160) * 2.0 ** -pw.wk_exp
    assert (flat[:, :147] - cw.reshape(10, 147)).abs().max().item() <= 2.0 ** -20 * cw.abs().max().item()
    assert flat[:, 147:].abs().max().item() == 0.0


def test_fully_connected_layer_cpu_path_unchanged():
    from invertavatar_amd.training.networks_stylegan2 import FullyConnectedLayer
    torch.manual_seed(3)
    fc = FullyConnectedLayer(32, 16, lr_multiplier=0.5, bias_init=1)
    x = torch.randn(3, 32)
    want = torch.addmm((fc.bias * fc.bias_gain).unsqueeze(0), x, (fc.weight * fc.weight_gain).t())
    with torch.no_grad():
        assert torch.allclose(fc(x), want, atol=1e-6)
