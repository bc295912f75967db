"""The SR head in its DEPLOYED precision (sr_num_fp16_res = 4, train_avatar_texture.py:215) against `sr_fp16.npz`, which
tests/golden/make_golden.py records from the REFERENCE head itself running its fp16 path (the `ws.device.type != 'cuda'` forcing of
training/networks_stylegan2.py:421-422 lifted in the fixture script only; VERDICT r5 hygiene 9b).  On CPU tensors the reference's ops are
its `_ref` implementations, whose bias_act rounds to fp16 after every step; the CUDA plugin rounds once.  The oracle restates both; the
product implements a third placement of the roundings (fp16 operands, fp32 accumulation and epilogue, one fp16 plane stored)."""
import numpy as np
import pytest
import torch

from invertavatar_amd import synthetic
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
from conftest import rnd, max_abs


def _head():
    g = TriPlaneGenerator(**synthetic.generator_kwargs('full', sr_num_fp16_res=4)).eval().requires_grad_(False)
    return synthetic.fill_parameters(g)


def _deviation(img, gld, tag):
    return max(max_abs(img[..., ::4, ::4], gld[f'{tag}_image_sub4']), max_abs(img[..., 224:288, 224:288], gld[f'{tag}_image_crop']),
               max_abs(torch.nn.functional.avg_pool2d(img.double(), 32).float(), gld[f'{tag}_image_block_means']))


def test_oracle_fp16_head_vs_reference_fp16_path(golden):
    """The restatement with per-step bias_act rounding reproduces the reference's fp16 head up to last-bit flips of the fp16-stored
    tensors (an fp16 ulp at |y| in [1, 2) is 9.8e-4): more than half of the pixels bit-equal, mean |d| < 1.5e-4, max within 2.5e-3.  The
    deployed (round-once) form -- what the CUDA plugins do, unpinnable here -- stays within 3e-3 of the same fixture."""
    from oracle import generator as OG
    gld = golden('sr_fp16.npz')
    g = _head()
    sd = {k[len('superresolution.'):]: v.detach() for k, v in g.state_dict().items() if k.startswith('superresolution.')}
    feat = rnd(31, 1, 32, 128, 128) * 0.5
    with torch.no_grad():
        ws = g.mapping(synthetic.latent(0, 1), synthetic.conditioning_camera(), truncation_psi=0.7, truncation_cutoff=14)
        assert max_abs(ws, gld['ws']) <= 1e-6
        per_op = OG.superresolution_8xdc_fp16(sd, feat[:, :3].contiguous(), feat, gld['ws'], per_op_bias_act=True).float()
        once = OG.superresolution_8xdc_fp16(sd, feat[:, :3].contiguous(), feat, gld['ws']).float()
    d = (per_op[..., 224:288, 224:288] - gld['fp16_image_crop']).abs()
    assert (d == 0).float().mean().item() >= 0.5 and d.mean().item() <= 1.5e-4, ((d == 0).float().mean().item(), d.mean().item())
    assert _deviation(per_op, gld, 'fp16') <= 2.5e-3
    assert _deviation(once, gld, 'fp16') <= 3e-3
    assert 1e-4 < max_abs(gld['fp16_image_sub4'], gld['fp32_image_sub4']) <= 3e-3          # the fixture's two modes differ: fp16 really ran


@pytest.mark.gpu
def test_product_fp16_head_vs_reference_fp16_path(golden):
    """The product's fp16 mode of the head on the fixture's seeded features: within TOL_RGB_FP16_SR of the reference's fp32 output and
    within TOL_RGB_FP16_SR_VS_RESTATEMENT of the reference's own fp16 output (tests/test_generator_gpu.py has the derivation of both)."""
    from invertavatar_amd.training import networks_stylegan2 as sg2
    from test_generator_gpu import TOL_RGB_FP16_SR, TOL_RGB_FP16_SR_VS_RESTATEMENT
    gld = golden('sr_fp16.npz')
    g = _head().cuda()
    feat = (rnd(31, 1, 32, 128, 128) * 0.5).cuda()
    saved, sg2.FP16_BLOCKS_COMPUTE_FP32 = sg2.FP16_BLOCKS_COMPUTE_FP32, False
    try:
        with torch.no_grad():
            img16 = g.superresolution(feat[:, :3].contiguous(), feat, gld['ws'].cuda(), noise_mode='none').float().cpu()
            sg2.FP16_BLOCKS_COMPUTE_FP32 = True
            img32 = g.superresolution(feat[:, :3].contiguous(), feat, gld['ws'].cuda(), noise_mode='none').float().cpu()
    finally:
        sg2.FP16_BLOCKS_COMPUTE_FP32 = saved
    d32, d16, d16_32 = _deviation(img32, gld, 'fp32'), _deviation(img16, gld, 'fp16'), _deviation(img16, gld, 'fp32')
    print(f'SR head on seeded features: fp32 mode vs reference fp32 {d32:.2e}; fp16 mode vs reference fp16 {d16:.2e}, vs reference fp32 {d16_32:.2e}')
    assert d32 <= 1e-4
    assert 1e-4 < d16_32 <= TOL_RGB_FP16_SR and d16 <= TOL_RGB_FP16_SR_VS_RESTATEMENT
