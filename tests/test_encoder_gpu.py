"""Few-shot inversion flow of eval_seq.py (encode -> 2 x AR_eval_forward with carried ConvGRU state -> drive frame)
on the GPU backend against the reference fixture (BASELINE configs 3/5 pattern, reduced to 8 source frames + 1 drive frame)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_few_shot_inversion_matches_reference(golden):
    from encoder_common import build_inversion_net, run_few_shot, compare_with_fixture
    net = build_inversion_net('full').cuda()
    ws, res, r_list, image = run_few_shot(net, 'cuda')
    worst = compare_with_fixture(golden('encoder_fewshot.npz'), ws, res, r_list, image, tol=2e-3)
    print(f'few-shot inversion: worst relative deviation {worst:.2e}')
    assert image.shape == (1, 3, 512, 512)
