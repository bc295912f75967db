"""Vector / box helpers (reference: volumetric_rendering/math_utils.py:18-118)."""
import torch


def transform_vectors(matrix, vectors4):
    """Left-multiply row vectors [N,M] by an MxM matrix."""
    return torch.matmul(vectors4, matrix.T)


def normalize_vecs(vectors):
    return vectors / torch.norm(vectors, dim=-1, keepdim=True)


def torch_dot(x, y):
    return (x * y).sum(-1)


def get_ray_limits_box(rays_o, rays_d, box_side_length):
    """Slab test against the axis-aligned cube of the given side centred at the origin.
    Returns (t_near, t_far) shaped like rays_o[..., :1]; both are -1 where the ray misses (math_utils.py:46-98)."""
    shape = rays_o.shape
    o = rays_o.detach().reshape(-1, 3)
    d = rays_d.detach().reshape(-1, 3)
    half = box_side_length / 2
    inv = 1.0 / d
    lo = (-half - o) * inv
    hi = (half - o) * inv
    t_near_axis = torch.minimum(lo, hi)
    t_far_axis = torch.maximum(lo, hi)
    # the reference chains the three slabs axis by axis and flags a miss whenever an interval is empty
    tmin, tmax = t_near_axis[:, 0], t_far_axis[:, 0]
    miss = torch.zeros_like(tmin, dtype=torch.bool)
    for ax in (1, 2):
        miss |= (tmin > t_far_axis[:, ax]) | (t_near_axis[:, ax] > tmax)
        tmin = torch.maximum(tmin, t_near_axis[:, ax])
        tmax = torch.minimum(tmax, t_far_axis[:, ax])
    tmin = torch.where(miss, torch.full_like(tmin, -1), tmin)
    tmax = torch.where(miss, torch.full_like(tmax, -2), tmax)
    return tmin.reshape(*shape[:-1], 1), tmax.reshape(*shape[:-1], 1)


def linspace(start, stop, num):
    """torch.linspace for tensor end points: result has `num` as a new leading dimension (math_utils.py:101-118)."""
    steps = torch.arange(num, dtype=torch.float32, device=start.device) / (num - 1)
    for _ in range(start.ndim):
        steps = steps.unsqueeze(-1)
    return start[None] + steps * (stop - start)[None]
