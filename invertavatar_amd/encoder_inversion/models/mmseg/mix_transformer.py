"""SegFormer "Mix Transformer" blocks used by the improved one-shot inversion encoders (reference:
encoder_inversion/models/mmseg/mix_transformer.py: ``Mlp`` :17, ``Attention`` :55, ``Block`` :118, ``OverlapPatchEmbed`` :155,
``MixVisionTransformer`` :193, ``DWConv`` :364, ``MLP`` :378, ``transformer_block`` :440).

Module and parameter names follow the reference so that its checkpoints load by name.  The reference pulls ``DropPath`` /
``to_2tuple`` / ``trunc_normal_`` from timm, which is not a dependency here: stochastic depth is inference-time identity, so
``DropPath`` is restated as the few lines it is.  Tokens are [B, N, C]; (H, W) travel beside them."""
import math

import torch
import torch.nn as nn

from ..layers import Conv2d

HIP_DWCONV = True      # Mix-FFN's depth-wise 3x3 (+ GELU) on the token grid through ia_dwconv3x3_tokens (device inference path)


HIP_ATTENTION = True      # device inference: softmax(QK^T)V of the 1024-dim / 4-head blocks through ia_attention
HIP_ATTENTION_SX_MIN = 16384      # N * M from which ia_attention_sx (fp16 pairs) replaces ia_attention (fp32 MFMAs): graph replay 24.4 vs 24.0 us at
                                  # 64 tokens, 32.3 vs 68.3 at 256


HIP_PATCH_EMBED = True      # device inference: OverlapPatchEmbed's strided convolution as ia_im2col_split + ia_linear_sx (tokens directly)
HIP_PATCH_EMBED_MIN_TOKENS = 1


HIP_LAYERNORM_SPLIT = True      # device inference: a block's LayerNorm inside the token split of the linear layers that read it


HIP_LINEAR = True      # device inference: the blocks' nn.Linear layers (q / kv / proj, fc1 / fc2) as fp16-pair GEMMs through ia_linear_sx


def _hip_linear_ok(x, *linears):
    return (HIP_LINEAR and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
            and all(m.in_features % 16 == 0 and m.weight.dtype == torch.float32 for m in linears))


def _split_normed(x, norm):
    """tokens_split(norm(x)): one launch when `norm` is a plain LayerNorm over a width ia_layernorm_split covers."""
    from .... import hipops
    if norm is None:
        return hipops.tokens_split(x.contiguous())
    if (HIP_LAYERNORM_SPLIT and isinstance(norm, nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None and len(norm.normalized_shape) == 1
            and hipops.layernorm_split_supported(x.shape[-1])):
        return hipops.layernorm_split(x.contiguous(), norm)
    return hipops.tokens_split(norm(x).contiguous())


def _hip_linear(lin, xs, gelu=False, residual=None):
    """nn.Linear on tokens already in the split format (hipops.tokens_split): weight split once per parameter version."""
    from .... import _runtime, hipops
    st = _runtime.state(lin)
    key = (lin.weight.data_ptr(), lin.weight._version, lin.weight.device)
    if getattr(st, 'wsplit_key', None) != key:
        st.wsplit, st.wsplit_key = hipops.pack_linear_weight_split(lin.weight), key
    return hipops.linear_sx(xs, st.wsplit, None if lin.bias is None else lin.bias.detach(), residual=residual, gelu=gelu)


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class DropPath(nn.Module):
    """Stochastic depth per sample (identity in eval mode / at rate 0)."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


def _init_weights(m):
    """Initialisation shared by every sub-module of the reference file (:30-44)."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            m.bias.data.zero_()


class DWConv(nn.Module):
    """3x3 depth-wise convolution on the token grid (positional information of the Mix-FFN)."""

    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)

    def forward(self, x, H, W, gelu=False, split=False):
        """`gelu`: also apply the GELU (erf form) that follows in Mix-FFN -- one launch on the device path (ia_dwconv3x3_tokens).
        `split`: return the result as hipops.SplitTokens, the operand of the linear layer behind it (device path only; None if not covered)."""
        B, N, C = x.shape
        conv = self.dwconv
        if (HIP_DWCONV and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and C % 4 == 0 and conv.kernel_size == (3, 3)
                and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.groups == C == conv.out_channels and conv.padding_mode == 'zeros'):
            from .... import _runtime, hipops
            st = _runtime.state(self)
            key = (conv.weight.data_ptr(), conv.weight._version, conv.weight.device)
            if getattr(st, 'w9c_key', None) != key:
                st.w9c, st.w9c_key = conv.weight.detach().float().reshape(C, 9).t().contiguous(), key
            bias = None if conv.bias is None else conv.bias.detach().float()
            if split:
                return hipops.dwconv3x3_tokens_split(x.contiguous(), st.w9c, bias, H, W, gelu=gelu) if C % 16 == 0 else None
            return hipops.dwconv3x3_tokens(x.contiguous(), st.w9c, bias, H, W, gelu=gelu)
        if split:
            return None
        y = conv(x.transpose(1, 2).reshape(B, C, H, W)).flatten(2).transpose(1, 2)
        return torch.nn.functional.gelu(y) if gelu else y


class Mlp(nn.Module):
    """Mix-FFN: fc1 -> depth-wise 3x3 -> GELU -> fc2."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        self.apply(_init_weights)

    def forward(self, x, H, W, residual=None, norm=None):
        """`residual`: added to the result (Block.forward's x + mlp(norm2(x)); in fc2's epilogue on the device path).  `norm`: a LayerNorm to
        apply to x first (Block.forward's norm2; on the device path inside the token split of fc1's input)."""
        exact_gelu = isinstance(self.act, nn.GELU) and getattr(self.act, 'approximate', 'none') == 'none'
        if exact_gelu and _hip_linear_ok(x, self.fc1, self.fc2) and not (self.training and self.drop.p > 0):
            from .... import hipops
            h1 = _hip_linear(self.fc1, _split_normed(x, norm))
            hs = self.dwconv(h1, H, W, gelu=True, split=True)          # convolution + GELU + the split of fc2's input in one launch
            if hs is None:
                hs = hipops.tokens_split(self.dwconv(h1, H, W, gelu=True).contiguous())
            return _hip_linear(self.fc2, hs, residual=None if residual is None else residual.contiguous())
        if norm is not None:
            x = norm(x)
        if exact_gelu:
            y = self.drop(self.fc2(self.drop(self.dwconv(self.fc1(x), H, W, gelu=True))))
        else:
            y = self.drop(self.fc2(self.drop(self.act(self.dwconv(self.fc1(x), H, W)))))
        return y if residual is None else residual + y


class Attention(nn.Module):
    """Multi-head self-attention with spatially reduced keys / values (sr_ratio > 1: a strided conv + LayerNorm on the token grid)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., sr_ratio=1):
        super().__init__()
        assert dim % num_heads == 0, f'dim {dim} should be divided by num_heads {num_heads}.'
        self.dim, self.num_heads = dim, num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.sr_ratio = sr_ratio
        if sr_ratio > 1:
            self.sr = Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)
        self.apply(_init_weights)

    def forward(self, x, H, W, residual=None, norm=None):
        """`residual`: added to the result (Block.forward's x + attn(norm1(x)); in proj's epilogue on the device path).  `norm`: a LayerNorm
        to apply to x first (Block.forward's norm1; with sr_ratio 1 on the device path inside the one token split that feeds q and kv)."""
        B, N, C = x.shape
        heads, hd = self.num_heads, C // self.num_heads
        lin = _hip_linear_ok(x, self.q, self.kv, self.proj) and not (self.training and self.proj_drop.p > 0)
        if lin and self.sr_ratio == 1:
            xs = _split_normed(x, norm)                        # one split for q and kv
        else:
            if norm is not None:
                x = norm(x)
            if lin:
                from .... import hipops
                xs = hipops.tokens_split(x.contiguous())
        if lin:
            from .... import hipops
            qp = _hip_linear(self.q, xs)
        else:
            qp = self.q(x)
        src = x
        if self.sr_ratio > 1:
            src = self.norm(self.sr(x.permute(0, 2, 1).reshape(B, C, H, W)).reshape(B, C, -1).permute(0, 2, 1))
        if lin:
            kv = _hip_linear(self.kv, xs if self.sr_ratio == 1 else hipops.tokens_split(src.contiguous()))
        else:
            kv = self.kv(src)

        def project(out):
            if lin:
                return _hip_linear(self.proj, hipops.tokens_split(out.contiguous()), residual=None if residual is None else residual.contiguous())
            y = self.proj_drop(self.proj(out))
            return y if residual is None else residual + y
        if (HIP_ATTENTION and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and not (self.training and self.attn_drop.p > 0)):
            from .... import _lib, hipops
            sx = HIP_LINEAR and hipops.attention_sx_supported(hd, N, kv.shape[1])
            if _lib.load().ia_attention_supported(hd, N, kv.shape[1]) and not (sx and hd == 256 and N * kv.shape[1] >= HIP_ATTENTION_SX_MIN):
                return project(hipops.attention(qp.contiguous(), kv.contiguous(), heads, self.scale))      # one launch, fp32 MFMAs: the 8^2 grids
            if sx:      # both products on fp16 pairs (head_dim 256: one launch, no score matrix): from 16^2 tokens up
                return project(hipops.attention_sx(qp.contiguous(), kv.contiguous(), heads, self.scale))
        q = qp.reshape(B, N, heads, hd).permute(0, 2, 1, 3)
        k, v = kv.reshape(B, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
        attn = self.attn_drop(((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1))
        return project((attn @ v).transpose(1, 2).reshape(B, N, C))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, sr_ratio=1):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop,
                              sr_ratio=sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.apply(_init_weights)

    def forward(self, x, H, W):
        if isinstance(self.drop_path, nn.Identity) or not self.training:      # (stochastic depth is the identity: the sums go into the projections)
            x = self.attn(x, H, W, residual=x, norm=self.norm1)
            return self.mlp(x, H, W, residual=x, norm=self.norm2)
        x = x + self.drop_path(self.attn(self.norm1(x), H, W))
        return x + self.drop_path(self.mlp(self.norm2(x), H, W))


class OverlapPatchEmbed(nn.Module):
    """Strided convolution with overlapping patches -> tokens + LayerNorm."""

    def __init__(self, img_size=224, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size, self.patch_size = _pair(img_size), _pair(patch_size)
        self.H, self.W = self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]
        self.num_patches = self.H * self.W
        self.proj = Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=stride,
                              padding=(self.patch_size[0] // 2, self.patch_size[1] // 2))
        self.norm = nn.LayerNorm(embed_dim)
        self.apply(_init_weights)

    def forward(self, x):
        conv = self.proj
        if (HIP_PATCH_EMBED and HIP_LINEAR and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and conv.kernel_size in ((7, 7), (3, 3)) and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
                and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros' and isinstance(conv.padding, tuple)):
            from .... import _runtime, hipops
            xs = hipops.im2col_split(x.contiguous(), conv.kernel_size[0], conv.stride[0], conv.padding[0])
            if xs.rows >= HIP_PATCH_EMBED_MIN_TOKENS:          # (few rows and a very long K: linear_sx_splitk cuts K over the launch)
                st = _runtime.state(conv)
                key = (conv.weight.data_ptr(), conv.weight._version, conv.weight.device)
                if getattr(st, 'patch_w_key', None) != key:
                    st.patch_w, st.patch_w_key = hipops.pack_patch_weight_split(conv.weight), key
                H, W = xs.grid
                tokens = hipops.linear_sx_splitk(xs, st.patch_w, None if conv.bias is None else conv.bias.detach())      # [B, H * W, C]: tokens directly
                return self.norm(tokens), H, W
        x = self.proj(x)
        H, W = x.shape[-2:]
        return self.norm(x.flatten(2).transpose(1, 2)), H, W


def _tokens_to_map(x, B, H, W):
    return x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()


class MixVisionTransformer(nn.Module):
    """Four-stage hierarchical encoder (patch_embed{1..4}, block{1..4}, norm{1..4}); returns the four feature maps."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dims=[64, 128, 256, 512],
                 num_heads=[1, 2, 4, 8], mlp_ratios=[4, 4, 4, 4], qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, depths=[3, 4, 6, 3], sr_ratios=[8, 4, 2, 1]):
        super().__init__()
        self.num_classes, self.depths = num_classes, depths
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        cur = 0
        for s in range(4):
            embed = OverlapPatchEmbed(img_size=img_size if s == 0 else img_size // (2 ** (s + 1)), patch_size=7 if s == 0 else 3,
                                      stride=4 if s == 0 else 2, in_chans=in_chans if s == 0 else embed_dims[s - 1], embed_dim=embed_dims[s])
            blocks = nn.ModuleList([Block(dim=embed_dims[s], num_heads=num_heads[s], mlp_ratio=mlp_ratios[s], qkv_bias=qkv_bias,
                                          qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[cur + i],
                                          norm_layer=norm_layer, sr_ratio=sr_ratios[s]) for i in range(depths[s])])
            setattr(self, f'patch_embed{s + 1}', embed)
            setattr(self, f'block{s + 1}', blocks)
            setattr(self, f'norm{s + 1}', norm_layer(embed_dims[s]))
            cur += depths[s]
        self.apply(_init_weights)

    def load_weights(self, pretrained=None, strict=False):
        """Copy the entries of a SegFormer checkpoint whose names and shapes match (:258-274)."""
        if not isinstance(pretrained, str):
            return
        own = self.state_dict()
        own.update({k: v for k, v in torch.load(pretrained, map_location='cpu').items() if k in own and (strict or v.shape == own[k].shape)})
        self.load_state_dict(own, strict=strict)

    def reset_drop_path(self, drop_path_rate):
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(self.depths))]
        cur = 0
        for s in range(4):
            for i, blk in enumerate(getattr(self, f'block{s + 1}')):
                blk.drop_path.drop_prob = dpr[cur + i]
            cur += self.depths[s]

    def forward_features(self, x):
        B, outs = x.shape[0], []
        for s in range(1, 5):
            x, H, W = getattr(self, f'patch_embed{s}')(x)
            for blk in getattr(self, f'block{s}'):
                x = blk(x, H, W)
            x = _tokens_to_map(getattr(self, f'norm{s}')(x), B, H, W)
            outs.append(x)
        return outs

    def forward(self, x):
        return self.forward_features(x)


class MLP(nn.Module):
    """Linear embedding of a feature map's channels."""

    def __init__(self, input_dim=2048, embed_dim=768):
        super().__init__()
        self.proj = nn.Linear(input_dim, embed_dim)

    def forward(self, x):
        t = x.flatten(2).transpose(1, 2)
        if _hip_linear_ok(t, self.proj):
            from .... import hipops
            return _hip_linear(self.proj, hipops.tokens_split(t.contiguous()))
        return self.proj(t)


class transformer_block(nn.Module):
    """Stride-2 overlap embedding to `embed_dim` tokens -> `num_vit` transformer blocks -> LayerNorm -> PixelShuffle back to the
    input resolution -> 1x1 conv to the input channel count (the refinement in front of each decoder stage, :440-458)."""

    def __init__(self, in_chans=256, embed_dim=1024, num_vit=2):
        super().__init__()
        self.patch_embed = OverlapPatchEmbed(img_size=0, stride=2, in_chans=in_chans, embed_dim=embed_dim)
        self.ViT = nn.ModuleList([Block(dim=embed_dim, num_heads=4, mlp_ratio=2, sr_ratio=1) for _ in range(num_vit)])
        self.pixel_shuffle = nn.PixelShuffle(upscale_factor=2)
        self.mlp = Conv2d(embed_dim // 4, in_chans, kernel_size=1)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, f):
        B = f.shape[0]
        f, H, W = self.patch_embed(f)
        for blk in self.ViT:
            f = blk(f, H, W)
        return self.mlp(self.pixel_shuffle(_tokens_to_map(self.norm(f), B, H, W)))
