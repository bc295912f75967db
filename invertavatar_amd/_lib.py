"""ctypes binding of libia_hip.so (declared in include/ia_hip.h).

The library is the product: if it cannot be loaded, every GPU op raises -- there is no
silent PyTorch fallback for device tensors.
"""
import ctypes
import os
import threading

import torch

from . import build as _build

_lock = threading.Lock()
_lib = None

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
_i64p = ctypes.POINTER(ctypes.c_int64)

ABI_VERSION = 7      # IA_HIP_ABI_VERSION of include/ia_hip.h

DTYPE_ID = {torch.float32: 0, torch.float16: 1, torch.float64: 2}

# name -> argtypes, mirroring include/ia_hip.h
_SIGNATURES = {
    'ia_version': [],
    'ia_last_error': [ctypes.c_char_p, ctypes.c_size_t],
    'ia_device_count': [],
    'ia_bias_act': [c_void_p] * 6 + [c_int, c_int64, c_int, c_int64, c_int, c_int, c_float, c_float, c_float, c_void_p],
    'ia_upfirdn2d': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, _i64p, c_int, c_int, _i64p,
                     c_int, c_int, _i64p] + [c_int] * 7 + [c_float, c_void_p],
    'ia_upfirdn2d_bias_act': [c_void_p] * 6 + [c_int] * 13 + [c_float, c_int, c_float, c_float, c_float, c_void_p],
    'ia_conv2d_mfma': [c_void_p] * 10 + [ctypes.c_size_t] + [c_int] * 8 + [c_float, c_float, c_float, c_int, c_void_p],
    'ia_conv2d_mfma_h': [c_void_p] * 10 + [ctypes.c_size_t] + [c_int] * 8 + [c_float, c_float, c_float, c_int, c_void_p],
    'ia_conv2d_mfma_s': [c_void_p] * 2 + [c_int] + [c_void_p] * 8 + [ctypes.c_size_t] + [c_int] * 8 + [c_float, c_float, c_float, c_int, c_void_p],
    'ia_conv2d_plan': [c_int] * 8 + [ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_size_t)],
    'ia_modconv_demod': [c_void_p] * 3 + [c_int] * 3 + [c_void_p],
    'ia_render_rays': [c_void_p] * 10 + [c_float, c_float, c_int] + [c_int] * 6 + [c_void_p] * 9 + [c_void_p, c_void_p, c_int] + [c_void_p],
    'ia_render_rays_grid': [c_int, c_int],
    'ia_render_rays_box': [c_void_p] * 6 + [ctypes.c_double, ctypes.c_double] + [c_void_p] * 4 + [c_float, c_float, c_int] + [c_int] * 6 + [c_void_p] * 9 + [c_void_p],
    'ia_ray_limits_box': [c_void_p, c_void_p, ctypes.c_double, c_int, c_int, c_void_p, c_void_p, c_void_p],
    'ia_ray_limits_box_parts': [c_int],
    'ia_importance_stage': [c_void_p] * 5 + [c_int, c_void_p],
    'ia_fill_mouth': [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'ia_mouth_edge_blur': [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'ia_conv2d_mfma_sx_rgb': [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 6 + [c_int] + [c_void_p] * 6 + [c_int, c_float] + [c_int] * 6 + [c_float] * 3 + [c_void_p],
    'ia_conv1x1': [c_void_p] * 6 + [c_int] * 5 + [c_float, c_void_p],
    'ia_torgb_supported': [c_int] * 5,
    'ia_torgb': [c_void_p] * 8 + [c_int] * 5 + [c_float, c_void_p],
    'ia_rasterize_level': [c_void_p] * 4 + [c_int64, c_void_p] + [c_int] * 9 + [c_void_p],
    'ia_blend_planes': [c_void_p] * 3 + [c_int64, c_void_p] + [c_int] * 5 + [c_void_p],
    'ia_channels_last': [c_void_p] * 2 + [c_int] * 4 + [c_void_p],
    'ia_cond_blend': [c_void_p] * 3 + [c_int] * 4 + [c_void_p],
    'ia_split_saturation_poll': [ctypes.POINTER(ctypes.c_uint), c_int, c_void_p],
    'ia_split_saturation_count': [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p],
    'ia_act_split': [c_void_p] * 4 + [c_int] * 5 + [c_void_p],
    'ia_conv2d_sx_supported': [c_int] * 6,
    'ia_conv2d_mfma_sx': [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 7 + [c_int] + [c_void_p] * 2 + [ctypes.c_size_t] + [c_int] * 7 + [c_float, c_void_p, c_float, c_float] + [c_int, c_void_p],
    'ia_bn_train_split': [c_void_p] * 7 + [c_int, c_void_p] + [c_int] * 5 + [c_float, c_float, c_void_p],
    'ia_upsample_bilinear_add': [c_void_p] * 3 + [c_int] * 5 + [c_void_p],
    'ia_dwconv3x3_tokens': [c_void_p] * 4 + [c_int] * 5 + [c_void_p],
    'ia_dwconv3x3_tokens_split': [c_void_p] * 4 + [c_int] * 5 + [c_void_p],
    'ia_conv3x3_s2_tiny_supported': [c_int] * 4,
    'ia_conv3x3_s2_tiny': [c_void_p] * 4 + [c_int] * 6 + [c_float, c_void_p],
    'ia_conv2d_down_plan': [c_int] * 5 + [ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_size_t)],
    'ia_conv2d_down_sx': [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 5 + [c_int, c_void_p, c_void_p, ctypes.c_size_t] + [c_int] * 6
                         + [c_float, c_void_p, c_float, c_float, c_int, c_void_p],
    'ia_upconv2d_rows_plan': [c_int] * 5 + [ctypes.POINTER(ctypes.c_size_t)],
    'ia_upconv2d_rows_sx': [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, ctypes.c_size_t] + [c_int] * 5 + [c_void_p],
    'ia_upconv2d_fir_sx': [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 5 + [c_void_p, c_int, c_void_p] + [c_int] * 6 + [c_float] * 3 + [c_void_p],
    'ia_fir_tail_split': [c_void_p] * 8 + [c_int] * 10 + [c_float, c_int, c_float, c_float, c_float, c_void_p],
    'ia_cond_blend_split': [c_void_p] * 4 + [c_int] * 4 + [c_void_p],
    'ia_filtered_lrelu': [c_void_p] * 5 + [c_int] * 17 + [c_float] * 3 + [c_int, c_void_p],
    'ia_convgru_gates': [c_void_p] * 4 + [c_int] * 4 + [c_void_p],
    'ia_convgru_gates_split': [c_void_p] * 4 + [c_int] * 4 + [c_void_p],
    'ia_convgru_update': [c_void_p] * 7 + [c_int] * 4 + [c_void_p],
    'ia_convgru_update_split': [c_void_p] * 7 + [c_int] * 4 + [c_void_p],
    'ia_se_gate': [c_void_p, _i64p, c_void_p, _i64p] + [c_void_p] * 4 + [c_int] * 5 + [c_void_p],
    'ia_se_gate_split': [c_void_p, _i64p, c_void_p, _i64p] + [c_void_p] * 7 + [c_int] * 5 + [c_void_p],
    'ia_attention_supported': [c_int] * 3,
    'ia_attention': [c_void_p] * 4 + [c_int] * 5 + [c_int64] * 8 + [c_float, c_void_p],
    'ia_tokens_split': [c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p],
    'ia_layernorm_split': [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p],
    'ia_tokens_split_t': [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p],
    'ia_attention_sx_supported': [c_int] * 3,
    'ia_attention_sx': [c_void_p] * 4 + [c_int] * 4 + [c_float, c_void_p],
    'ia_softmax_split': [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'ia_matmul_sx': [c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_int64, c_int64, c_int, c_int64, c_int64, c_int64, c_int64, c_float, c_void_p],
    'ia_im2col_split': [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p],
    'ia_linear_splitk_plan': [c_int, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_size_t)],
    'ia_linear_sx_splitk': [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, ctypes.c_size_t, c_void_p],
    'ia_linear_sx': [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p],
    'ia_uv_rasterize': [c_void_p] * 5 + [c_int] * 8 + [c_float, c_int, c_void_p],
    'ia_layout_grid_u8': [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p],
    'ia_stage_inputs': [ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), _i64p, c_int, c_void_p],
    'ia_ray_sampler': [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p],
    'ia_styles_demod': [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p],
}


def declared_symbols():
    """Entry points parsed from include/ia_hip.h (used by the CPU-side export test)."""
    import re
    hdr = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'include', 'ia_hip.h')
    with open(hdr) as fh:
        text = fh.read()
    return sorted(set(re.findall(r'^\s*(?:int|size_t)\s+(ia_[a-z0-9_]+)\s*\(', text, flags=re.M)))


def load():
    """Load (building it first if the .so is absent) and type the C-ABI library."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = os.environ.get('IA_HIP_LIB')          # explicit library (kernel experiments); otherwise the in-tree build
        if not path:
            path = _build.LIB_PATH
            if not os.path.exists(path):
                path = _build.build()
        lib = ctypes.CDLL(path)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_size_t if name == 'ia_last_error' else c_int
        if lib.ia_version() != ABI_VERSION:
            raise RuntimeError(f'libia_hip.so ABI version {lib.ia_version()} != {ABI_VERSION}; rebuild with python -m invertavatar_amd.build')
        _lib = lib
        return lib


def last_error():
    buf = ctypes.create_string_buffer(512)
    load().ia_last_error(buf, 512)
    return buf.value.decode()


def check(status, what):
    if status != 0:
        raise RuntimeError(f'{what}: {last_error()} (ia_status {status})')


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream, as an integer for ctypes."""
    return torch.cuda.current_stream(device).cuda_stream


def strides64(t):
    return (ctypes.c_int64 * t.ndim)(*t.stride())
