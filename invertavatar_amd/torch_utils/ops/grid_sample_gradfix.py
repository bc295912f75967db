"""grid_sample entry point (reference: torch_utils/ops/grid_sample_gradfix.py:28-31): with
`enabled = False` it is a passthrough to bilinear / zeros / align_corners=False sampling."""
import torch

enabled = False


def grid_sample(input, grid):
    return torch.nn.functional.grid_sample(input=input, grid=grid, mode='bilinear', padding_mode='zeros',
                                           align_corners=False)
