"""Output side of the generator path: float frames -> uint8 picture grid (reference: ``layout_grid`` in
reenact_avatar_next3d.py:117-131, eval_seq.py and the other video scripts).

Device fp32 batches go through ONE HIP pass (``ia_layout_grid_u8``: scale, clamp, truncate, tile, NCHW -> HWC); everything else
(CPU tensors, uint8 inputs, odd shapes) follows the torch formulation.  ``to_uint8_hwc`` is the per-frame form that is
all-gathered between GPUs (a quarter of the fp32 bytes) and handed to a video writer."""
import torch

from . import _lib


def _grid_u8_hip(img, grid_w, grid_h, chw_to_hwc):
    b, c, h, w = img.shape
    shape = (grid_h * h, grid_w * w, c) if chw_to_hwc else (c, grid_h * h, grid_w * w)
    out = torch.empty(shape, dtype=torch.uint8, device=img.device)
    with torch.cuda.device(img.device):
        st = _lib.load().ia_layout_grid_u8(img.data_ptr(), out.data_ptr(), b, c, h, w, grid_w, grid_h, int(chw_to_hwc),
                                           _lib.stream_ptr(img.device))
    _lib.check(st, 'ia_layout_grid_u8')
    return out


def layout_grid(img, grid_w=None, grid_h=1, float_to_uint8=True, chw_to_hwc=True, to_numpy=True):
    """Same signature and result as the reference helper: [B,C,H,W] -> one [grid_h*H, grid_w*W, C] picture."""
    batch_size, channels, img_h, img_w = img.shape
    if grid_w is None:
        grid_w = batch_size // grid_h
    assert batch_size == grid_w * grid_h
    if (float_to_uint8 and img.is_cuda and img.dtype == torch.float32 and img_w % 4 == 0 and channels in (1, 3, 4)
            and not torch.is_grad_enabled()):
        out = _grid_u8_hip(img.contiguous(), grid_w, grid_h, chw_to_hwc)
    else:
        if float_to_uint8:
            img = (img * 127.5 + 128).clamp(0, 255).to(torch.uint8)
        out = img.reshape(grid_h, grid_w, channels, img_h, img_w).permute(2, 0, 3, 1, 4).reshape(channels, grid_h * img_h, grid_w * img_w)
        if chw_to_hwc:
            out = out.permute(1, 2, 0)
    return out.cpu().numpy() if to_numpy else out


def to_uint8_hwc(img):
    """[B,3,H,W] float frames in [-1,1] -> [B,H,W,3] uint8 (reenact_avatar_next3d.py:123,127 per frame)."""
    b, c, h, w = img.shape
    if img.is_cuda and img.dtype == torch.float32 and w % 4 == 0 and c in (1, 3, 4):
        return _grid_u8_hip(img.contiguous(), 1, b, True).view(b, h, w, c)
    return (img * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
