"""Oracle restatement of the Next3D++ generator forward pass (SURVEY.md 8a rows
N2-N8, G1-G3, S1).  Works on a flat ``{reference parameter name: tensor}`` dict
(names per SURVEY.md C14) so it shares no code with the product's nn.Modules.
Test infrastructure only.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops, renderer

BBOX_256 = (57, 185, 64, 192)  # triplane_v20.py:114


def sub(sd, prefix):
    """View of the entries below `prefix.` with the prefix stripped."""
    n = len(prefix) + 1
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix + '.')}


def mapping(sd, z, c, num_ws, truncation_psi=1.0, truncation_cutoff=None, lr_mul=0.01, num_layers=2):
    """MappingNetwork.forward.  Reference: networks_stylegan2_new.py:233-268."""
    def norm2(v):
        return v * (v.square().mean(dim=1, keepdim=True) + 1e-8).rsqrt()
    x = norm2(z.float())
    y = norm2(ops.fully_connected(c.float(), sd['embed.weight'], sd['embed.bias']))
    x = torch.cat([x, y], 1)
    for i in range(num_layers):
        x = ops.fully_connected(x, sd[f'fc{i}.weight'], sd[f'fc{i}.bias'], lr_mul, 'lrelu')
    x = x.unsqueeze(1).repeat(1, num_ws, 1)
    if truncation_psi != 1:
        cut = num_ws if truncation_cutoff is None else truncation_cutoff
        x[:, :cut] = sd['w_avg'].lerp(x[:, :cut], truncation_psi)
    return x


def synthesis_layer(sd, x, w, up, noise_mode, fused=True, gain=1.0, conv_clamp=None):
    """SynthesisLayer.forward.  Reference: training/networks_stylegan2.py:311-330."""
    styles = ops.fully_connected(w, sd['affine.weight'], sd['affine.bias'])
    noise = None
    if noise_mode == 'const':
        noise = sd['noise_const'] * sd['noise_strength']
    y = ops.modulated_conv2d(x, sd['weight'], styles, noise=noise, up=up, padding=1,
                             resample_filter=sd['resample_filter'], fused=fused)
    clamp = conv_clamp * gain if conv_clamp is not None else None
    return ops.bias_act(y, sd['bias'], act='lrelu', gain=ops.SQRT2 * gain, clamp=clamp)


def to_rgb(sd, x, w, fused=True, conv_clamp=None):
    """ToRGBLayer.forward.  Reference: training/networks_stylegan2.py:351-357."""
    in_ch = sd['weight'].shape[1]
    styles = ops.fully_connected(w, sd['affine.weight'], sd['affine.bias']) * (1 / math.sqrt(in_ch))
    y = ops.modulated_conv2d(x, sd['weight'], styles, demodulate=False, fused=fused)
    return ops.bias_act(y, sd['bias'], clamp=conv_clamp)


def synthesis_block(sd, x, img, ws, condition=None, noise_mode='const', fused=True, conv_clamp=None):
    """SynthesisBlock.forward, architecture 'skip', fp32.
    Reference: networks_stylegan2_new.py:417-467 (SR: training/networks_stylegan2.py:417-460)."""
    wi = 0
    if 'const' in sd:
        x = sd['const'].unsqueeze(0).repeat(ws.shape[0], 1, 1, 1)
        x = synthesis_layer(sub(sd, 'conv1'), x, ws[:, wi], 1, noise_mode, fused, conv_clamp=conv_clamp); wi += 1
    else:
        x = synthesis_layer(sub(sd, 'conv0'), x, ws[:, wi], 2, noise_mode, fused, conv_clamp=conv_clamp); wi += 1
        if condition is not None:  # CS-SFT, :448-452
            half = x.shape[1] // 2
            x = torch.cat([x[:, :half], x[:, half:] * condition[0] + condition[1]], 1)
        x = synthesis_layer(sub(sd, 'conv1'), x, ws[:, wi], 1, noise_mode, fused, conv_clamp=conv_clamp); wi += 1
    if img is not None:
        img = ops.upsample2d(img, sd['resample_filter'])
    y = to_rgb(sub(sd, 'torgb'), x, ws[:, wi], fused, conv_clamp)
    img = img + y if img is not None else y
    return x, img


def synthesis_network(sd, ws, cond_list=None, return_list=False, feat_conditions=None, img_resolution=256,
                      out_res=(32, 256), noise_mode='const', fused=True):
    """SynthesisNetwork.forward.  Reference: networks_stylegan2_new.py:509-548."""
    res_list = [2 ** i for i in range(2, int(math.log2(img_resolution)) + 1)]
    x = img = None
    feats = []
    start = int(math.log2(out_res[0])) - 2
    end = int(math.log2(out_res[1])) - 2
    w_idx = 0
    for idx, res in enumerate(res_list):
        bsd = sub(sd, f'b{res}')
        n_conv = 1 if res == 4 else 2
        cur = ws[:, w_idx:w_idx + n_conv + 1]
        w_idx += n_conv
        cond = feat_conditions.get(res) if feat_conditions is not None else None
        x, img = synthesis_block(bsd, x, img, cur, cond, noise_mode, fused)
        if idx >= start:
            if return_list:
                if idx == start:
                    feats.append(img.clone())
                feats.append(x.clone())
            if cond_list is not None:
                if idx == start:
                    a = cond_list[0][:, -1:]
                    img = cond_list[0][:, :-1] * a + img * (1 - a)
                if idx < end:
                    cimg = cond_list[1 + idx - start]
                    a = cimg[:, -1:]
                    x = cimg[:, :-1] * a + x * (1 - a)
    if return_list:
        feats.append(img)
        return feats
    return img


def rasterize(texture_feats, uvcoords_image, static_feats, bbox_256=BBOX_256):
    """TriPlaneGenerator.rasterize.  Reference: triplane_v20.py:317-339."""
    uv = uvcoords_image.float()
    grid, alpha = uv[..., :2], uv[..., 2:].permute(0, 3, 1, 2)
    full_alpha, mouth = renderer.fill_mouth(alpha.clone())
    upper = mouth.clone()
    upper[:, :, :87] = 0
    upper_alpha = torch.clamp(alpha + upper, 0, 1)
    outs = []
    for tex, sta in zip(texture_feats, static_feats):
        res = tex.shape[2]
        bb = [round(i * res / 256) for i in bbox_256]
        rend = ops.resize_bilinear_aa(ops.grid_sample_bilinear(tex, grid), (res, res))
        a = ops.resize_bilinear_aa(alpha, (res, res))
        s = ops.resize_bilinear_aa(sta[:, :, bb[0]:bb[1], bb[2]:bb[3]], (res, res))
        outs.append(torch.cat([rend * a + s * (1 - a), ops.resize_bilinear_aa(upper_alpha, (res, res))], 1))
    return outs, full_alpha, mouth


def blend_planes(stitch, full_alpha, static_plane, bbox_256=BBOX_256):
    """Paste the 128^2-resized stitch into plane 0 and alpha-blend.
    Reference: triplane_v20.py:119-128."""
    y0, y1, x0, x1 = bbox_256
    s_canvas = torch.zeros_like(stitch)
    a_canvas = torch.zeros_like(full_alpha)
    s_canvas[:, :, y0:y1, x0:x1] = ops.resize_bilinear_aa(stitch, (128, 128))
    a_canvas[:, :, y0:y1, x0:x1] = ops.resize_bilinear_aa(full_alpha, (128, 128))
    planes = static_plane.clone()
    planes[:, 0] = s_canvas * a_canvas + static_plane[:, 0] * (1 - a_canvas)
    return planes


def superresolution_8xdc(sd, rgb, x, ws, noise_mode='none', fused=True, conv_clamp=None):
    """SuperresolutionHybrid8XDC.forward, fp32.  Reference: superresolution.py:278-289."""
    ws3 = ws[:, -1:, :].repeat(1, 3, 1)
    if x.shape[-1] != 128:
        x = ops.resize_bilinear_aa(x, (128, 128))
        rgb = ops.resize_bilinear_aa(rgb, (128, 128))
    x, rgb = synthesis_block(sub(sd, 'block0'), x, rgb, ws3, None, noise_mode, fused, conv_clamp)
    x, rgb = synthesis_block(sub(sd, 'block1'), x, rgb, ws3, None, noise_mode, fused, conv_clamp)
    return rgb


def _h(t):
    """One fp16 storage rounding (the tensor lives in fp16 in the reference's fp16 blocks)."""
    return t.to(torch.float16).to(torch.float32)


def _modconv_fp16(x, weight, styles, up, resample_filter, demodulate):
    """modulated_conv2d with x.dtype == float16, fused branch (training/networks_stylegan2.py:52-57,60-66,81-91): weight and styles
    are pre-normalised by their max-norms when demodulating (:54-56), the per-sample weight w * s * d is formed in fp32 and ROUNDED
    to fp16 (`w.to(x.dtype)`, :87), the convolution runs on fp16 operands (fp32 accumulation in the library) and every tensor it
    writes is fp16: the conv output, and for up = 2 the stride-2 transposed conv output AND the FIR output
    (conv2d_resample.py:114-131).  Pinned since r06 by tests/golden/sr_fp16.npz: the reference head itself run in fp16 on CPU tensors
    with its fp32 forcing (networks_stylegan2.py:421-422) lifted in the fixture script; see superresolution_8xdc_fp16(per_op_bias_act)."""
    o, i, kh, kw = weight.shape
    if demodulate:
        weight = weight * (1 / math.sqrt(i * kh * kw) / weight.abs().amax(dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.abs().amax(dim=1, keepdim=True)
    outs = []
    for b in range(x.shape[0]):
        wb = weight * styles[b].reshape(1, i, 1, 1)
        if demodulate:
            wb = wb * (wb.square().sum(dim=[1, 2, 3], keepdim=True) + 1e-8).rsqrt()
        wb = _h(wb)
        if up == 1:
            yb = _h(F.conv2d(x[b:b + 1], wb, padding=kh // 2))
        else:
            yb = _h(F.conv_transpose2d(x[b:b + 1], wb.transpose(0, 1), stride=2))
            yb = _h(ops.upfirdn2d(yb, resample_filter, padding=(1, 1, 1, 1), gain=4.0))
        outs.append(yb)
    return torch.cat(outs, 0)


def _bias_act_fp16(y, b, act, gain, clamp, per_op):
    """bias_act on an fp16 tensor.  per_op=False: computed in fp32 and rounded once on store, as the CUDA plugin does (bias_act.cu:19-22,
    149) -- the deployed path.  per_op=True: the reference's `_bias_act_ref` (bias_act.py:93-122), which is what runs when the fp16 blocks
    are executed on CPU tensors: every step (add bias, activation, gain) is a torch op on the fp16 tensor, i.e. rounds to fp16."""
    if not per_op:
        return _h(ops.bias_act(y, _h(b), act=act, gain=gain, clamp=clamp))
    x = _h(y + _h(b).reshape(1, -1, 1, 1))
    if act == 'lrelu':
        x = _h(F.leaky_relu(x, 0.2))
    if gain != 1:
        x = _h(x * gain)
    return x.clamp(-clamp, clamp) if clamp is not None else x


def synthesis_block_fp16(sd, x, img, ws, conv_clamp=256, per_op_bias_act=False):
    """SynthesisBlock.forward with use_fp16 on a CUDA device, noise_mode 'none' (training/networks_stylegan2.py:417-460): x is cast
    to fp16 at the top (:436-437); each SynthesisLayer = fp16 modulated conv -> bias_act computed in fp32, stored fp16, clamp 256
    (:327-329); ToRGB likewise (no demodulation: no pre-normalisation) and its result goes to fp32 before the skip add (:456-458)."""
    x = _h(x)
    wi = 0
    for name, up in (('conv0', 2), ('conv1', 1)):
        lsd = sub(sd, name)
        styles = ops.fully_connected(ws[:, wi], lsd['affine.weight'], lsd['affine.bias']); wi += 1
        y = _modconv_fp16(x, lsd['weight'], styles, up, lsd['resample_filter'], True)
        x = _bias_act_fp16(y, lsd['bias'], 'lrelu', ops.SQRT2, conv_clamp, per_op_bias_act)
    img = ops.upsample2d(img, sd['resample_filter'])
    tsd = sub(sd, 'torgb')
    styles = ops.fully_connected(ws[:, wi], tsd['affine.weight'], tsd['affine.bias']) * (1 / math.sqrt(tsd['weight'].shape[1]))
    y = _modconv_fp16(x, tsd['weight'], styles, 1, None, False)
    y = _bias_act_fp16(y, tsd['bias'], 'linear', 1, conv_clamp, per_op_bias_act)
    return x, img + y


def superresolution_8xdc_fp16(sd, rgb, x, ws, per_op_bias_act=False):
    """SuperresolutionHybrid8XDC with sr_num_fp16_res > 0 as deployed (superresolution.py:263-289, use_fp16 + conv_clamp 256).
    per_op_bias_act=True: the same head as the reference runs it on CPU tensors once its fp32 forcing is lifted (`_bias_act_ref` rounds after
    every step) -- the form tests/golden/sr_fp16.npz records from the reference itself (make_golden.py:gen_sr_fp16), which pins everything
    else of this restatement: the pre-normalisation, the fp16 rounding of w * s * d, the fp16 storage of every tensor, the clamp."""
    ws3 = ws[:, -1:, :].repeat(1, 3, 1)
    if x.shape[-1] != 128:
        x = ops.resize_bilinear_aa(x, (128, 128))
        rgb = ops.resize_bilinear_aa(rgb, (128, 128))
    x, rgb = synthesis_block_fp16(sub(sd, 'block0'), x, rgb, ws3, per_op_bias_act=per_op_bias_act)
    x, rgb = synthesis_block_fp16(sub(sd, 'block1'), x, rgb, ws3, per_op_bias_act=per_op_bias_act)
    return rgb


def split_static(static_feats):
    """triplane_v20.py:109-112: keep plane 0 of the two 96-channel entries."""
    b = static_feats[0].shape[0]
    plane = static_feats[-1].view(b, 3, 32, *static_feats[-1].shape[-2:])
    out = list(static_feats)
    out[0] = static_feats[0].view(b, 3, 32, *static_feats[0].shape[-2:])[:, 0]
    out[-1] = plane[:, 0]
    return out, plane


def synthesis(sd, ws, c, uvcoords_image, jitter, nrr=128, texture_feats=None, static_feats=None, fused=True,
              return_all=False):
    """TriPlaneGenerator.synthesis / synthesis_withTexture, evaluation=True, noise_mode='const'.
    Reference: triplane_v20.py:89-150 and :152-244."""
    cam = c[:, -25:]
    c2w = cam[:, :16].reshape(-1, 4, 4)
    k = cam[:, 16:25].reshape(-1, 3, 3)
    rays_o, rays_d = renderer.ray_sampler_zxc(c2w, k, nrr)
    if texture_feats is None:
        texture_feats = synthesis_network(sub(sd, 'texture_backbone.synthesis'), ws, return_list=True, fused=fused)
    if static_feats is None:
        static_feats = synthesis_network(sub(sd, 'backbone.synthesis'), ws, return_list=True, fused=fused)
    static_for_raster, static_plane = split_static(static_feats)
    cond, full_alpha, mouth = rasterize(texture_feats, uvcoords_image, static_for_raster)
    stitch = synthesis_network(sub(sd, 'face_backbone.synthesis'), ws, cond_list=cond, fused=fused)
    planes = blend_planes(stitch, full_alpha, static_plane)
    dec = sub(sd, 'decoder')
    feat, depth, wsum = renderer.render(planes, dec, rays_o, rays_d, jitter)
    b = ws.shape[0]
    feature_image = feat.permute(0, 2, 1).reshape(b, feat.shape[-1], nrr, nrr).contiguous()
    depth_image = depth.permute(0, 2, 1).reshape(b, 1, nrr, nrr)
    rgb = feature_image[:, :3]
    image = superresolution_8xdc(sub(sd, 'superresolution'), rgb, feature_image, ws)
    out = dict(image=image, image_raw=rgb, image_depth=depth_image)
    if return_all:
        out.update(feature_image=feature_image, triplane=planes, texture=texture_feats, static=static_feats,
                   cond=cond, full_alpha=full_alpha, stitch=stitch)
    return out


def blended_planes(sd, ws, uvcoords_image, texture_feats=None, static_feats=None, static_feat_conditions=None, fused=True):
    """The tri-planes every entry point of TriPlaneGenerator builds before rendering (triplane_v20.py:262-290 = :356-380)."""
    if texture_feats is None:
        texture_feats = synthesis_network(sub(sd, 'texture_backbone.synthesis'), ws, return_list=True, fused=fused)
    if static_feats is None:
        static_feats = synthesis_network(sub(sd, 'backbone.synthesis'), ws, return_list=True, feat_conditions=static_feat_conditions,
                                         fused=fused)
    static_for_raster, static_plane = split_static(static_feats)
    cond, full_alpha, _ = rasterize(texture_feats, uvcoords_image, static_for_raster)
    stitch = synthesis_network(sub(sd, 'face_backbone.synthesis'), ws, cond_list=cond, fused=fused)
    return blend_planes(stitch, full_alpha, static_plane), texture_feats, static_feats


def synthesis_with_condition(sd, ws, c, uvcoords_image, jitter, nrr, static_feat_conditions=None):
    """TriPlaneGenerator.synthesis_withCondition with noise_mode='const' (=> evaluation, :293).  Reference: triplane_v20.py:246-315."""
    cam = c[:, -25:]
    rays_o, rays_d = renderer.ray_sampler_zxc(cam[:, :16].reshape(-1, 4, 4), cam[:, 16:25].reshape(-1, 3, 3), nrr)
    planes, texture_feats, static_feats = blended_planes(sd, ws, uvcoords_image, static_feat_conditions=static_feat_conditions)
    feat, depth, _ = renderer.render(planes, sub(sd, 'decoder'), rays_o, rays_d, jitter)
    b = ws.shape[0]
    feature_image = feat.permute(0, 2, 1).reshape(b, feat.shape[-1], nrr, nrr).contiguous()
    image = superresolution_8xdc(sub(sd, 'superresolution'), feature_image[:, :3], feature_image, ws)
    return dict(image=image, image_raw=feature_image[:, :3], image_depth=depth.permute(0, 2, 1).reshape(b, 1, nrr, nrr),
                feature_image=feature_image, triplane=planes, static=static_feats, texture=texture_feats)


def query_points(sd, ws, coordinates, uvcoords_image, box_warp=1.0):
    """TriPlaneGenerator.sample_mixed (and sample, after the mapping network): density + colour features at arbitrary points.
    Reference: triplane_v20.py:341-402 -> renderer.py:353-363."""
    planes, _, _ = blended_planes(sd, ws, uvcoords_image)
    rgb, sigma = renderer.osg_decoder(sub(sd, 'decoder'), renderer.sample_from_planes(planes, coordinates, box_warp))
    return dict(rgb=rgb, sigma=sigma)
