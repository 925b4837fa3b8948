"""Camera with the matrix conventions of FluidDynamics/scene/camera.py:84-110 (all 4x4 stored
transposed, i.e. for row-vector x matrix).  Data loading is out of scope;
this class only builds what the render pipes and kernels consume."""
from __future__ import annotations

import numpy as np
import torch

from ..utils.graphics_utils import get_projection_matrix, get_projection_matrix_cv, get_world_2_view2


class Camera:
    def __init__(self, R, T, FoVx, FoVy, image_width, image_height, image=None, uid=0, trans=np.array([0.0, 0.0, 0.0]),
                 scale=1.0, cxr=0.0, cyr=0.0, device="cuda", image_name=""):
        self.uid = uid
        self.R = np.asarray(R, dtype=np.float64)
        self.T = np.asarray(T, dtype=np.float64)
        self.FoVx = float(FoVx)
        self.FoVy = float(FoVy)
        self.image_name = image_name
        self.image_width = int(image_width)
        self.image_height = int(image_height)
        self.original_image = None if image is None else image.clamp(0.0, 1.0).to(device)
        self.z_far = 100.0   # camera.py:84
        self.z_near = 0.01   # camera.py:85
        self.trans = trans
        self.scale = scale
        dev = torch.device(device)
        self.data_device = dev
        self.world_view_transform = torch.tensor(get_world_2_view2(self.R, self.T, trans, scale)).transpose(0, 1).to(dev)
        if cyr != 0.0:
            proj = get_projection_matrix_cv(self.z_near, self.z_far, self.FoVx, self.FoVy, cx=cxr, cy=cyr)
        else:
            proj = get_projection_matrix(self.z_near, self.z_far, self.FoVx, self.FoVy)
        self.projection_matrix = proj.transpose(0, 1).to(dev)
        self.full_proj_transform = self.world_view_transform.unsqueeze(0).bmm(
            self.projection_matrix.unsqueeze(0)).squeeze(0)
        self.camera_center = self.world_view_transform.inverse()[3, :3]


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """(R, T) in the reference's convention: W2C = [R^T | T], camera x right, y down, z forward."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    f = target - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, up)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    R = np.stack([r, d, f], axis=1)
    T = -R.T @ eye
    return R, T
