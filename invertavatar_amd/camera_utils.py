"""Top-level ``camera_utils`` of the reference (identical to training_avatar_texture/camera_utils.py there)."""
from .training_avatar_texture.camera_utils import *  # noqa: F401,F403
from .training_avatar_texture.camera_utils import (GaussianCameraPoseSampler, LookAtPoseSampler, UniformCameraPoseSampler,  # noqa: F401
                                                  create_cam2world_matrix, FOV_to_intrinsics)
