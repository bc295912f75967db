"""The output kernel (ia_layout_grid_u8) and the H1 harness on the device."""
import numpy as np
import pytest
import torch

from invertavatar_amd import output
from conftest import rnd
from test_harness_cpu import run_harness, check_mosaics

pytestmark = pytest.mark.gpu


def _ref_grid(img, grid_w, grid_h, hwc):
    b, c, h, w = img.shape
    u8 = (img * 127.5 + 128).clamp(0, 255).to(torch.uint8)
    out = u8.reshape(grid_h, grid_w, c, h, w).permute(2, 0, 3, 1, 4).reshape(c, grid_h * h, grid_w * w)
    return out.permute(1, 2, 0) if hwc else out


def test_layout_grid_kernel_is_bit_exact(golden):
    g = golden('harness.npz')
    x = g['grid_in'].cuda()
    assert np.array_equal(output.layout_grid(x, grid_w=3, grid_h=2), g['grid_3x2_hwc'].numpy())
    assert np.array_equal(output.layout_grid(x, grid_w=None, grid_h=1, chw_to_hwc=False), g['grid_6x1_chw'].numpy())
    assert np.array_equal(output.layout_grid(x, grid_w=1, grid_h=6), g['grid_1x6_hwc'].numpy())
    # values on and around every integer boundary, out-of-range values, all channel counts, full frame size
    for c, (b, gw, gh) in ((3, (4, 2, 2)), (1, (2, 2, 1)), (4, (3, 1, 3))):
        img = rnd(5 + c, b, c, 64, 96) * 1.5
        ramp = (torch.arange(-20, 276, dtype=torch.float32) - 128) / 127.5
        img.view(-1)[:ramp.numel()] = ramp
        img.view(-1)[ramp.numel():2 * ramp.numel()] = torch.nextafter(ramp, torch.tensor(10.0))
        for hwc in (True, False):
            got = output.layout_grid(img.cuda(), grid_w=gw, grid_h=gh, chw_to_hwc=hwc, to_numpy=False).cpu()
            assert torch.equal(got, _ref_grid(img, gw, gh, hwc).contiguous()), (c, hwc)
    big = rnd(9, 8, 3, 512, 512)
    assert torch.equal(output.to_uint8_hwc(big.cuda()).cpu(), output.to_uint8_hwc(big))


def test_reenactment_loop_matches_the_script_on_device(golden):
    gld = golden('harness.npz')
    check_mosaics(gld, *run_harness(gld, 'cuda'))
