"""Camera pose samplers and intrinsics (reference: camera_utils.py = training_avatar_texture/camera_utils.py).

Conventions: y up, z forward, x left; `horizontal_mean` is the azimuth (rotation about y), `vertical_mean` the polar
angle from +y; a camera on the +z axis has azimuth pi/2... exactly as the reference's docstrings state
(camera_utils.py:23-37)."""
import math

import torch

from .volumetric_rendering import math_utils


def _origins_on_sphere(h, v, radius):
    v = torch.clamp(v, 1e-5, math.pi - 1e-5)
    theta = h
    phi = torch.arccos(1 - 2 * (v / math.pi))
    o = torch.zeros((h.shape[0], 3), device=h.device)
    o[:, 0:1] = radius * torch.sin(phi) * torch.cos(math.pi - theta)
    o[:, 2:3] = radius * torch.sin(phi) * torch.sin(math.pi - theta)
    o[:, 1:2] = radius * torch.cos(phi)
    return o


class GaussianCameraPoseSampler:
    """Yaw/pitch ~ N(mean, stddev); camera looks at the origin."""

    @staticmethod
    def sample(horizontal_mean, vertical_mean, horizontal_stddev=0, vertical_stddev=0, radius=1, batch_size=1, device='cpu'):
        h = torch.randn((batch_size, 1), device=device) * horizontal_stddev + horizontal_mean
        v = torch.randn((batch_size, 1), device=device) * vertical_stddev + vertical_mean
        o = _origins_on_sphere(h, v, radius)
        return create_cam2world_matrix(math_utils.normalize_vecs(-o), o)


class LookAtPoseSampler:
    """As above but looking at `lookat_position`."""

    @staticmethod
    def sample(horizontal_mean, vertical_mean, lookat_position, horizontal_stddev=0, vertical_stddev=0, radius=1, batch_size=1,
               device='cpu'):
        h = torch.randn((batch_size, 1), device=device) * horizontal_stddev + horizontal_mean
        v = torch.randn((batch_size, 1), device=device) * vertical_stddev + vertical_mean
        o = _origins_on_sphere(h, v, radius)
        return create_cam2world_matrix(math_utils.normalize_vecs(lookat_position - o), o)


class UniformCameraPoseSampler:
    """Yaw/pitch uniform in mean +- stddev; camera looks at the origin."""

    @staticmethod
    def sample(horizontal_mean, vertical_mean, horizontal_stddev=0, vertical_stddev=0, radius=1, batch_size=1, device='cpu'):
        h = (torch.rand((batch_size, 1), device=device) * 2 - 1) * horizontal_stddev + horizontal_mean
        v = (torch.rand((batch_size, 1), device=device) * 2 - 1) * vertical_stddev + vertical_mean
        o = _origins_on_sphere(h, v, radius)
        return create_cam2world_matrix(math_utils.normalize_vecs(-o), o)


def create_cam2world_matrix(forward_vector, origin):
    """cam2world from a viewing direction and a position; y is up, no roll (camera_utils.py:118-137)."""
    fwd = math_utils.normalize_vecs(forward_vector)
    up = torch.tensor([0, 1, 0], dtype=torch.float, device=origin.device).expand_as(fwd)
    right = -math_utils.normalize_vecs(torch.cross(up, fwd, dim=-1))
    up = math_utils.normalize_vecs(torch.cross(fwd, right, dim=-1))
    n = fwd.shape[0]
    rot = torch.eye(4, device=origin.device).unsqueeze(0).repeat(n, 1, 1)
    rot[:, :3, :3] = torch.stack((right, up, fwd), dim=-1)
    trans = torch.eye(4, device=origin.device).unsqueeze(0).repeat(n, 1, 1)
    trans[:, :3, 3] = origin
    cam2world = trans @ rot
    assert cam2world.shape[1:] == (4, 4)
    return cam2world


def FOV_to_intrinsics(fov_degrees, device='cpu'):
    """Normalised 3x3 intrinsics; the reference's constants (3.14159, 1.414) are kept (camera_utils.py:140-148)."""
    focal = float(1 / (math.tan(fov_degrees * 3.14159 / 360) * 1.414))
    return torch.tensor([[focal, 0, 0.5], [0, focal, 0.5], [0, 0, 1]], device=device)
