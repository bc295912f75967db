"""Synthetic, seed-determined inputs for the generator path.

There are no datasets or checkpoints on the bench box, so weights, latents, cameras,
UV-mesh conditions and the stratified-sampling jitter are all functions of small integer
seeds.  NumPy ``RandomState`` is used throughout because it is portable across machines.
The golden-fixture generator, the parity tests and ``bench.py`` all draw from here, which is
what makes "the same inputs" well defined between the reference, the oracle and the HIP path.
"""
import math
import zlib

import numpy as np
import torch

SR_MODULE = 'training_avatar_texture.superresolution.SuperresolutionHybrid8XDC'


def rendering_kwargs(**overrides):
    """The rendering options the reference trains and infers with
    (train_avatar_texture.py:320-348, 355, 365-367)."""
    kw = dict(superresolution_module=SR_MODULE, sr_antialias=True, superresolution_noise_mode='none',
              c_gen_conditioning_zero=False, c_scale=1, decoder_lr_mul=1, depth_resolution=48,
              depth_resolution_importance=48, disparity_space_sampling=False, clamp_mode='softplus', box_warp=1,
              ray_start=2.25, ray_end=3.3, avg_camera_radius=2.7, avg_camera_pivot=[0, 0, 0.2])
    kw.update(overrides)
    return kw


def generator_kwargs(width='full', sr_num_fp16_res=0):
    """Constructor kwargs of TriPlaneGenerator.  width='full' is the BASELINE model
    (channel_base 32768, channel_max 512: 88.3 M parameters); 'small' keeps the topology and
    resolutions but caps the backbones at 32 channels (used by golden fixtures)."""
    base, cmax = (32768, 512) if width == 'full' else (2048, 32)
    return dict(z_dim=512, c_dim=25, w_dim=512, img_resolution=512, img_channels=3, sr_num_fp16_res=sr_num_fp16_res,
                mapping_kwargs=dict(num_layers=2), channel_base=base, channel_max=cmax,
                fused_modconv_default='inference_only', num_fp16_res=0, conv_clamp=None,
                sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default='inference_only'),
                rendering_kwargs=rendering_kwargs())


def _seed(name, salt=0):
    return (zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF


def fill_parameters(module_or_state, salt=0):
    """Overwrite every parameter/buffer with a value that depends only on its NAME and shape.

    Independent of construction order, so the reference model, the oracle's flat dict and the
    product's modules all receive bit-identical weights.  Distributions follow the reference's
    initialisation (randn weights, affine bias 1) but biases / noise strengths are made non-zero so
    that those code paths are exercised.  FIR filters are left untouched."""
    state = module_or_state.state_dict() if hasattr(module_or_state, 'state_dict') else module_or_state
    with torch.no_grad():
        for name in sorted(state.keys()):
            t = state[name]
            leaf = name.rsplit('.', 1)[-1]
            if leaf == 'resample_filter' or not t.dtype.is_floating_point:
                continue
            rs = np.random.RandomState(_seed(name, salt))
            v = rs.randn(*t.shape).astype(np.float32) if t.ndim else np.float32(rs.randn())
            if leaf == 'bias':
                v = v * 0.1 + (1.0 if '.affine.' in name else 0.0)
            elif leaf == 'noise_strength':
                v = v * 0.1
            elif leaf == 'w_avg':
                v = v * 0.1
            t.copy_(torch.as_tensor(v, dtype=t.dtype).reshape(t.shape))
    return module_or_state


def latent(seed, batch=1):
    """z as the reenactment script draws it (reenact_avatar_next3d.py:172)."""
    return torch.from_numpy(np.random.RandomState(seed).randn(batch, 512)).float()


def _normalize(v):
    return v / np.linalg.norm(v)


def look_at_pose(yaw, pitch, pivot=(0.0, 0.0, 0.2), radius=2.7):
    """4x4 cam2world of a camera on a sphere of `radius` about `pivot`... restated in float64
    NumPy from the look-at construction of camera_utils.py:58-80,118-137."""
    pitch = min(max(pitch, 1e-5), math.pi - 1e-5)
    phi = math.acos(1 - 2 * (pitch / math.pi))
    origin = np.array([radius * math.sin(phi) * math.cos(math.pi - yaw), radius * math.cos(phi),
                       radius * math.sin(phi) * math.sin(math.pi - yaw)])
    fwd = _normalize(np.asarray(pivot, dtype=np.float64) - origin)
    right = -_normalize(np.cross([0.0, 1.0, 0.0], fwd))
    up = _normalize(np.cross(fwd, right))
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, fwd, origin
    return m


def intrinsics(fov_degrees=18.837):
    """Normalised pinhole intrinsics (camera_utils.py:140-148; note the reference's 3.14159)."""
    focal = 1.0 / (math.tan(fov_degrees * 3.14159 / 360.0) * 1.414)
    return np.array([[focal, 0, 0.5], [0, focal, 0.5], [0, 0, 1.0]])


def camera_label(frame, n_frames=240):
    """25-float label for frame k of the yaw/pitch orbit of eval_updated_os.py:214-220."""
    ang = 2 * math.pi * frame / n_frames
    c2w = look_at_pose(math.pi / 2 + 0.35 * math.sin(ang), math.pi / 2 - 0.05 + 0.25 * math.cos(ang))
    return torch.from_numpy(np.concatenate([c2w.reshape(-1), intrinsics().reshape(-1)]).astype(np.float32))


def camera_labels(frames):
    return torch.stack([camera_label(k) for k in frames], 0)


def conditioning_camera():
    """Frontal camera used for the mapping network (reenact_avatar_next3d.py:174-177)."""
    c2w = look_at_pose(math.pi / 2, math.pi / 2)
    return torch.from_numpy(np.concatenate([c2w.reshape(-1), intrinsics().reshape(-1)]).astype(np.float32))[None]


def uv_condition(frame, n_frames=240, res=256):
    """uvcoords_image [res,res,3]: u = x, v = y on a [-1,1] grid; mask = disc with a rectangular
    mouth hole whose height follows the frame."""
    lin = np.linspace(-1, 1, res)
    ys, xs = np.meshgrid(lin, lin, indexing='ij')
    mask = ((xs ** 2 + ys ** 2) < 0.5).astype(np.float32)
    half = 10 + int(round(8 * math.sin(2 * math.pi * frame / n_frames)))
    mask[160 - half:160 + half, 100:156] = 0
    return torch.from_numpy(np.stack([xs, ys, mask], -1).astype(np.float32))


def uv_conditions(frames):
    return torch.stack([uv_condition(k) for k in frames], 0)


def jitter(frames, n_rays, n_coarse=48):
    """Stratified-sampling jitter in [0,1): the tensor that replaces torch.rand_like at
    volumetric_rendering/renderer.py:406.  One RandomState(1234 + k) stream per frame k."""
    out = [np.random.RandomState(1234 + k).rand(n_rays, n_coarse, 1).astype(np.float32) for k in frames]
    return torch.from_numpy(np.stack(out, 0))


def fill_encoder_parameters(net, skip_prefix='generator.', salt=0):
    """Name-seeded parameters for the plain torch.nn encoders (IR-SE50 trunks, UNets, style heads).

    Unlike the generator's equalised-lr layers these have no built-in weight gain, so weights are drawn with a
    1/sqrt(fan_in) scale to keep 50 residual units numerically sane; BatchNorm statistics are made non-trivial."""
    def draw(name, shape, scale=1.0, shift=0.0, absolute=False):
        rs = np.random.RandomState(_seed(name, salt))
        v = rs.randn(*shape).astype(np.float32) if len(shape) else np.float32(rs.randn())
        if absolute:
            v = np.abs(v)
        return torch.as_tensor(v * scale + shift, dtype=torch.float32).reshape(shape)

    with torch.no_grad():
        for mname, mod in net.named_modules():
            if mname.startswith(skip_prefix) or mname == skip_prefix.rstrip('.'):
                continue
            kind = type(mod).__name__
            own = dict(mod.named_parameters(recurse=False))
            own.update(dict(mod.named_buffers(recurse=False)))
            for pname, t in own.items():
                full = f'{mname}.{pname}' if mname else pname
                if not t.dtype.is_floating_point or full in ('latent_avg', 'black_uv_bg'):
                    continue
                shape = tuple(t.shape)
                if kind == 'FullyConnectedLayer':
                    t.copy_(draw(full, shape) if pname == 'weight' else draw(full, shape, 0.1))
                elif kind == 'BatchNorm2d':
                    if pname == 'weight': t.copy_(draw(full, shape, 0.1, 1.0))
                    elif pname == 'running_var': t.copy_(draw(full, shape, 0.1, 1.0, absolute=True))
                    else: t.copy_(draw(full, shape, 0.1))
                elif kind == 'LayerNorm':
                    t.copy_(draw(full, shape, 0.1, 1.0) if pname == 'weight' else draw(full, shape, 0.1))
                elif kind == 'PReLU':
                    t.copy_(draw(full, shape, 0.05, 0.25))
                elif pname == 'weight' and t.ndim >= 2:
                    fan_in = int(np.prod(shape[1:]))
                    t.copy_(draw(full, shape, 1.0 / math.sqrt(fan_in)))
                else:
                    t.copy_(draw(full, shape, 0.1))
        if hasattr(net, 'latent_avg'):
            net.latent_avg.copy_(net.generator.backbone.mapping.w_avg.reshape(1, 512))
    return net


def source_frames(seed, n, res=512):
    """n smooth random RGB frames in [-1, 1] (the "real" source frames of the few-shot inversion)."""
    rs = np.random.RandomState(seed)
    low = torch.from_numpy(rs.rand(n, 3, res // 16, res // 16).astype(np.float32)) * 2 - 1
    return torch.nn.functional.interpolate(low, size=(res, res), mode='bilinear', align_corners=False)


def source_uv(seed, frames):
    """x['uv'] [n,6,256,256]: 3 channels of ground-truth texture in UV space + (u, v, mask)."""
    rs = np.random.RandomState(seed)
    low = torch.from_numpy(rs.rand(len(frames), 3, 16, 16).astype(np.float32)) * 2 - 1
    gttex = torch.nn.functional.interpolate(low, size=(256, 256), mode='bilinear', align_corners=False)
    pverts = uv_conditions(frames).permute(0, 3, 1, 2)
    return torch.cat([gttex, pverts], 1)
