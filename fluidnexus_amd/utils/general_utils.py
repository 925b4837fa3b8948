"""Small helpers with the semantics of FluidDynamics/utils/general_utils.py (inv_sigmoid :10-11,
get_expon_lr_func :63-94, build_rotation :113-158, build_scaling_rotation :182-191, safe_state
:194-217).  Device follows the input tensor instead of the reference's hard-coded "cuda"."""
from __future__ import annotations

import random

import numpy as np
import torch


def inv_sigmoid(x):
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation lr_init -> lr_final over max_steps, optionally eased in over
    lr_delay_steps by a sine ramp starting at lr_delay_mult."""
    log0, log1 = (np.log(lr_init), np.log(lr_final)) if lr_init > 0 and lr_final > 0 else (0.0, 0.0)

    def lr_at(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        ease = 1.0
        if lr_delay_steps > 0:
            ease = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return ease * np.exp(log0 * (1 - t) + log1 * t)

    return lr_at


def build_rotation(r):
    """Quaternion (w, x, y, z) rows -> rotation matrices; normalises first."""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def build_scaling_rotation(s, r):
    return build_rotation(r) @ torch.diag_embed(s)


def strip_symmetric(sym):
    return torch.stack([sym[:, 0, 0], sym[:, 0, 1], sym[:, 0, 2], sym[:, 1, 1], sym[:, 1, 2], sym[:, 2, 2]], dim=1)


def safe_state(silent=True, device="cuda:0"):
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    if torch.cuda.is_available() and str(device).startswith("cuda"):
        torch.cuda.set_device(torch.device(device))
