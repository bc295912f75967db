"""ia_tokens_split / ia_linear_sx (csrc/linear_split.hip): the transformer blocks' nn.Linear layers as fp16-pair GEMMs, against torch fp64
(reference: encoder_inversion/models/mmseg/mix_transformer.py:18-116 -- F.linear + bias, GELU, the residual sums of Block.forward)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(m_shape, k, n, seed):
    from conftest import rnd
    x = (rnd(seed, *m_shape, k) * 10.0 ** (rnd(seed + 1, *m_shape, k) * 1.5 - 1.0).clamp(-5, 2)).cuda()
    w = (rnd(seed + 2, n, k) * 0.02).cuda()
    return x, w, rnd(seed + 3, n).cuda(), rnd(seed + 4, *m_shape, n).cuda()


# the three tile forms: 128 x 128 workgroup tiles (>= 256 of them), 64 x 64, one 32 x 64 tile with K over the four waves; ragged M / N
@pytest.mark.parametrize('m_shape,k,n', [((1, 4096), 1024, 4096), ((2, 1024), 1024, 1024), ((1, 64), 1024, 2048), ((1, 4096), 2048, 1024),
                                         ((3, 37), 48, 72), ((1, 1000), 64, 200), ((1, 2100), 32, 2100)])
def test_linear_on_fp16_pairs_matches_fp64(m_shape, k, n):
    from invertavatar_amd import hipops
    x, w, bias, res = _case(m_shape, k, n, 50)
    xs = hipops.tokens_split(x)
    ws = hipops.pack_linear_weight_split(w)
    ref = x.double() @ w.double().t()
    bound = x.double().abs() @ w.double().abs().t()

    parts = bound + bias.double().abs() + res.double().abs()        # magnitudes the fp32 sums of the epilogue round at

    def close(got, want, what):
        ratio = ((got.double() - want).abs() / (5e-7 * bound + 2e-7 * parts + 1e-9)).max().item()
        assert got.shape == want.shape and ratio <= 1.0, (what, ratio)
    hipops.split_saturation_poll()
    close(hipops.linear_sx(xs, ws), ref, 'bare')
    close(hipops.linear_sx(xs, ws, bias), ref + bias.double(), 'bias')
    close(hipops.linear_sx(xs, ws, bias, residual=res), ref + bias.double() + res.double(), 'bias + residual')
    g = torch.nn.functional.gelu(ref + bias.double())
    got = hipops.linear_sx(xs, ws, bias, residual=res, gelu=True)
    # (GELU's slope is <= 1.13: the pre-activation bound carries over; erff itself is good to a few ulp)
    ratio = ((got.double() - (g + res.double())).abs() / (6e-7 * bound + 4e-7 * parts + 1e-7)).max().item()
    assert ratio <= 1.0, ratio
    assert not hipops.split_saturation_poll()
    assert torch.equal(hipops.linear_sx(xs, ws, bias), hipops.linear_sx(xs, ws, bias))        # fixed summation order


def test_tokens_split_is_the_split_format():
    """hi + lo * 2^-11 reproduces the tokens to 2^-22; values outside the fp16 range saturate and raise the range-watch word."""
    from conftest import rnd
    from invertavatar_amd import hipops
    x = (rnd(60, 2, 50, 48) * 10.0 ** (rnd(61, 2, 50, 48) * 2 - 1).clamp(-6, 3)).cuda()
    hipops.split_saturation_poll()
    xs = hipops.tokens_split(x)
    assert xs.data.shape == (2, 6, 100, 8) and xs.rows == 100 and xs.cols == 48 and xs.lead_shape == (2, 50)
    back = (xs.data[0].float() + xs.data[1].float() / 2048.0).permute(1, 0, 2).reshape(100, 48)
    # (|v| < 2^-14 rides entirely in the low part: 11 bits of a value that small, <= 1.5e-8 absolute)
    assert ((back - x.reshape(100, 48)).abs() <= 2.0 ** -21 * x.reshape(100, 48).abs() + 1.5e-8).all()
    assert not hipops.split_saturation_poll()
    x[1, 3, 5] = 7e4
    hipops.tokens_split(x)
    assert hipops.split_saturation_poll()
    with pytest.raises(RuntimeError, match='ia_tokens_split needs'):
        hipops.tokens_split(torch.zeros(4, 40, device='cuda'))


def test_transformer_block_on_hip_linears_matches_torch():
    """Block.forward with the linear layers, the attention and the depth-wise convolution on the device kernels against the same module
    with every switch off (ATen / rocBLAS), at the flow's dimensions (1 024 dims, 4 heads, mlp_ratio 2) on a 16 x 16 token grid."""
    from invertavatar_amd.encoder_inversion.models.mmseg import mix_transformer as mt
    from conftest import rnd
    torch.manual_seed(0)
    blk = mt.Block(dim=1024, num_heads=4, mlp_ratio=2, sr_ratio=1).cuda().eval()
    for p in blk.parameters():
        if p.dim() == 2:
            p.data.normal_(0, 0.03)
    x = rnd(70, 1, 256, 1024).cuda()
    with torch.no_grad():
        got = blk(x, 16, 16)
        saved = mt.HIP_LINEAR, mt.HIP_ATTENTION, mt.HIP_DWCONV
        try:
            mt.HIP_LINEAR = mt.HIP_ATTENTION = mt.HIP_DWCONV = False
            ref = blk.double()(x.double(), 16, 16)
        finally:
            mt.HIP_LINEAR, mt.HIP_ATTENTION, mt.HIP_DWCONV = saved
    assert (got.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('b,c,h,ks,stride,n', [(1, 128, 64, 7, 2, 256), (2, 24, 40, 7, 4, 96), (1, 5, 33, 3, 2, 40), (1, 224, 32, 7, 2, 1024)])
def test_patch_embedding_as_im2col_free_gemm(b, c, h, ks, stride, n):
    """OverlapPatchEmbed's strided convolution (mix_transformer.py:155-190) through ia_im2col_split + ia_linear_sx: tokens [B, OH * OW, N]
    against conv2d in fp64, flattened the reference's way; channel counts whose K = C * ks^2 is not a multiple of 16 are zero-padded."""
    from conftest import rnd
    from invertavatar_amd import hipops
    x = rnd(80, b, c, h, h + 6).cuda()
    w = (rnd(81, n, c, ks, ks) / (c * ks * ks) ** 0.5).cuda()
    bias = rnd(82, n).cuda()
    xs = hipops.im2col_split(x, ks, stride, ks // 2)
    got = hipops.linear_sx(xs, hipops.pack_patch_weight_split(w), bias)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), stride=stride, padding=ks // 2)
    assert xs.grid == tuple(ref.shape[-2:])
    ref = ref.flatten(2).transpose(1, 2)
    assert got.shape == ref.shape and (got.double() - ref).abs().max().item() <= 3e-6 * max(1.0, ref.abs().max().item())


def test_overlap_patch_embed_routes():
    from invertavatar_amd.encoder_inversion.models.mmseg import mix_transformer as mt
    from conftest import rnd
    torch.manual_seed(1)
    pe = mt.OverlapPatchEmbed(img_size=0, stride=2, in_chans=128, embed_dim=256).cuda().eval()
    x = rnd(83, 1, 128, 48, 48).cuda()
    with torch.no_grad():
        got, H, W = pe(x)
        saved = mt.HIP_PATCH_EMBED
        try:
            mt.HIP_PATCH_EMBED = False
            ref, H2, W2 = pe(x)
        finally:
            mt.HIP_PATCH_EMBED = saved
    assert (H, W) == (H2, W2) == (24, 24) and got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize('one_launch', [True, False])
@pytest.mark.parametrize('b,n,m,c,heads', [(1, 1024, 1024, 1024, 4), (2, 100, 48, 64, 2), (1, 77, 4096, 512, 2), (1, 300, 160, 256, 1), (2, 130, 48, 512, 2)])
def test_attention_on_the_fp16_pair_gemm(b, n, m, c, heads, one_launch, monkeypatch):
    """hipops.attention_sx (ia_tokens_split / _t, ia_matmul_sx, ia_softmax_split) against Attention.forward's arithmetic in fp64
    (mix_transformer.py:83-116: q @ k^T * scale, softmax over the keys, @ v, heads back into the token layout); head_dim 256 in one launch
    (ia_attention_sx: no score matrix) or, with the switch off and for other head sizes, as matmul / softmax / matmul."""
    from conftest import rnd
    from invertavatar_amd import hipops
    monkeypatch.setattr(hipops, 'ATTENTION_SX_ONE_LAUNCH', one_launch)
    q, kv = rnd(90, b, n, c).cuda(), rnd(91, b, m, 2 * c).cuda()
    hd = c // heads
    scale = hd ** -0.5
    got = hipops.attention_sx(q, kv, heads, scale)
    qd = q.double().reshape(b, n, heads, hd).permute(0, 2, 1, 3)
    k, v = kv.double().reshape(b, m, 2, heads, hd).permute(2, 0, 3, 1, 4)
    ref = (((qd @ k.transpose(-2, -1)) * scale).softmax(dim=-1) @ v).transpose(1, 2).reshape(b, n, c)
    assert got.shape == ref.shape and (got.double() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())


def test_attention_module_routes_large_grids_through_the_gemm():
    from invertavatar_amd.encoder_inversion.models.mmseg import mix_transformer as mt
    from conftest import rnd
    torch.manual_seed(2)
    att = mt.Attention(1024, num_heads=4, sr_ratio=1).cuda().eval()
    for p in att.parameters():
        if p.dim() == 2:
            p.data.normal_(0, 0.03)
    x = rnd(92, 1, 1024, 1024).cuda()
    with torch.no_grad():
        got = att(x, 32, 32)
        saved = mt.HIP_LINEAR, mt.HIP_ATTENTION
        try:
            mt.HIP_LINEAR = mt.HIP_ATTENTION = False
            ref = att.double()(x.double(), 32, 32)
        finally:
            mt.HIP_LINEAR, mt.HIP_ATTENTION = saved
    assert (got.double() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())


def test_linear_with_k_cut_over_the_launch():
    """Few rows x a very long K (the deepest patch embedding): ia_linear_splitk_plan cuts K, slice products are summed in slice order."""
    import ctypes
    from conftest import rnd
    from invertavatar_amd import _lib, hipops
    m, k, n = 64, 50176, 256
    x, w, bias = rnd(95, 1, m, k).cuda(), (rnd(96, n, k) * 0.01).cuda(), rnd(97, n).cuda()
    ks, nbytes = ctypes.c_int(0), ctypes.c_size_t(0)
    _lib.check(_lib.load().ia_linear_splitk_plan(m, k, n, ctypes.byref(ks), ctypes.byref(nbytes)), 'plan')
    assert ks.value > 1 and nbytes.value == ks.value * m * n * 4 and k % (16 * ks.value) == 0
    xs, ws = hipops.tokens_split(x), hipops.pack_linear_weight_split(w)
    got = hipops.linear_sx_splitk(xs, ws, bias)
    ref = x.double() @ w.double().t() + bias.double()
    bound = x.double().abs() @ w.double().abs().t()
    assert got.shape == ref.shape and ((got.double() - ref).abs() <= 5e-7 * bound + 1e-6).all()
    assert torch.equal(got, hipops.linear_sx_splitk(xs, ws, bias))
    _lib.check(_lib.load().ia_linear_splitk_plan(4096, 1024, 1024, ctypes.byref(ks), ctypes.byref(nbytes)), 'plan')
    assert ks.value == 1 and nbytes.value == 0


@pytest.mark.parametrize('m_shape,k', [((1, 4096), 1024), ((2, 37), 512), ((1, 5), 2048)])
def test_layernorm_inside_the_token_split(m_shape, k):
    """ia_layernorm_split = ia_tokens_split(nn.LayerNorm(x)) in one launch: the reconstructed values against LayerNorm in fp64."""
    from conftest import rnd
    from invertavatar_amd import hipops
    x = (rnd(100, *m_shape, k) * 3 + 0.7).cuda()
    norm = torch.nn.LayerNorm(k).cuda()
    norm.weight.data = (rnd(101, k) * 0.2 + 1).cuda()
    norm.bias.data = (rnd(102, k) * 0.1).cuda()
    xs = hipops.layernorm_split(x, norm)
    m = x.numel() // k
    assert xs.data.shape == (2, k // 8, m, 8) and xs.lead_shape == tuple(m_shape)
    back = (xs.data[0].float() + xs.data[1].float() / 2048.0).permute(1, 0, 2).reshape(m, k)
    ref = torch.nn.functional.layer_norm(x.double(), (k,), norm.weight.double(), norm.bias.double(), norm.eps).reshape(m, k)
    assert (back.double() - ref).abs().max().item() <= 3e-6
    with pytest.raises(RuntimeError, match='ia_layernorm_split covers'):
        hipops.layernorm_split(torch.zeros(4, 768, device='cuda'), torch.nn.LayerNorm(768).cuda())


def test_dwconv_writes_the_split_of_its_result():
    from conftest import rnd
    from invertavatar_amd import hipops
    b, h, w, c = 2, 9, 12, 48
    x, w9c, bias = rnd(110, b, h * w, c).cuda(), (rnd(111, 9, c) * 0.3).cuda(), rnd(112, c).cuda()
    for gelu in (False, True):
        want = hipops.tokens_split(hipops.dwconv3x3_tokens(x, w9c, bias, h, w, gelu=gelu))
        got = hipops.dwconv3x3_tokens_split(x, w9c, bias, h, w, gelu=gelu)
        assert got.rows == want.rows and got.cols == want.cols and torch.equal(got.data, want.data)
