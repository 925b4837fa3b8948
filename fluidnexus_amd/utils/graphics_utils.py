"""Camera-matrix helpers with the conventions of FluidDynamics/utils/graphics_utils.py
(get_world_2_view2 :24-35, get_projection_matrix :38-60, get_projection_matrix_cv :101-147,
fov2focal/focal2fov).  Pinned by tests/golden/graphics_utils.npz (generated from the reference)."""
from __future__ import annotations

import math

import numpy as np
import torch


def get_world_2_view(R, t):
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    return np.float32(Rt)


def get_world_2_view2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """World->view with the camera centre moved by `translate` then scaled (done in fp64)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).T
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)
    c2w[:3, 3] = (c2w[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(c2w))


def _frustum_matrix(z_near, z_far, left, right, bottom, top):
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * z_near / (right - left)
    P[1, 1] = 2.0 * z_near / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = (z_far + z_near) / (z_far - z_near)  # reference keeps (f+n)/(f-n), graphics_utils.py:57
    P[2, 3] = -(z_far * z_near) / (z_far - z_near)
    return P


def get_projection_matrix(z_near, z_far, fovX, fovY):
    top = math.tan(fovY / 2) * z_near
    right = math.tan(fovX / 2) * z_near
    return _frustum_matrix(z_near, z_far, -right, right, -top, top)


def get_projection_matrix_cv(z_near, z_far, fovX, fovY, cx=0.0, cy=0.0):
    """Off-centre principal point; cx, cy in [-0.5, 0.5] as fractions of the image size."""
    tx, ty = math.tan(fovX / 2), math.tan(fovY / 2)
    top, right = ty * z_near, tx * z_near
    dx, dy = (2 * tx * z_near) * cx, (2 * ty * z_near) * cy
    return _frustum_matrix(z_near, z_far, -right + dx, right + dx, -top + dy, top + dy)


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))
