"""Seeded synthetic Gaussians / cameras / particles for the BASELINE.json configurations
(SURVEY.md section 8(d) table).  Everything is generated on the CPU with numpy so that the HIP
path and the CPU oracle see bit-identical inputs."""
from __future__ import annotations

import math

import numpy as np
import torch

from .scene.camera import Camera, look_at


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def random_gaussians(P, seed=0, box=0.5, log_scale=(-5.5, -3.5), channels=3, center=(0.0, 0.0, 0.0)):
    """config 1 cloud: xyz ~ U(box), log-scale ~ U(lo,hi), quat ~ N(0,1) normalised, opacity = sigmoid(N)."""
    rng = np.random.RandomState(seed)
    xyz = (rng.uniform(-box, box, size=(P, 3)) + np.asarray(center)).astype(np.float32)
    scales = np.exp(rng.uniform(log_scale[0], log_scale[1], size=(P, 3))).astype(np.float32)
    q = rng.normal(size=(P, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    opacity = _sigmoid(rng.normal(size=(P, 1))).astype(np.float32)
    colors = rng.uniform(0, 1, size=(P, channels)).astype(np.float32)
    return dict(means3D=xyz, scales=scales, rotations=q, opacities=opacity, colors=colors)


def plume_gaussians(P, seed=0, center=(0.34, 0.0, -0.225), radius=0.1, y_range=(-0.02, 0.6), log_scale=-5.9,
                    opacity=0.1, grey=0.7, channels=1):
    """config 2 'ScalarReal-like' fluid cloud: cylinder around `center`, constant scale e^-5.9
    (gm_dynamics.py:172), opacity 0.1 (:173), grey 0.7 (:171)."""
    rng = np.random.RandomState(seed)
    ang = rng.uniform(0, 2 * math.pi, size=P)
    rad = radius * np.sqrt(rng.uniform(0, 1, size=P))
    xyz = np.stack([center[0] + rad * np.cos(ang), rng.uniform(y_range[0], y_range[1], size=P),
                    center[2] + rad * np.sin(ang)], axis=1).astype(np.float32)
    scales = np.full((P, 3), math.exp(log_scale), np.float32)
    q = np.zeros((P, 4), np.float32)
    q[:, 0] = 1.0
    return dict(means3D=xyz, scales=scales, rotations=q, opacities=np.full((P, 1), opacity, np.float32),
                colors=np.full((P, channels), grey, np.float32))


PLUME_TARGET = (0.34, 0.3, -0.225)  # what the benchmark cameras look at


def backdrop_gaussians(P, seed=0, ring=False, channels=3, target=PLUME_TARGET, log_scale=(-5.0, -3.0), occluding=False):
    """Static background Gaussians of configs 3/4/5 (log-scale U(-5,-3), opacity sigmoid(N), RGB U(0,1)) placed
    BEHIND the plume as seen from the benchmark cameras, like the room behind the smoke in the reference's captures:
      arc cameras (z > target z, 120 degree arc): a slab 0.4 .. 1.0 m behind the plume, 3 m wide, 1.6 m high;
      ring cameras (full circle at 1.6 m): a cylindrical wall of radius 2.2 .. 2.6 m around the plume axis -- every
      camera sees the far side of it behind the plume, the near side is behind the camera.
    (Round 1 scattered them in a box AROUND the plume: the opaque cloud hid the plume from every camera, so the
    fluid's image gradient was identically zero -- a benchmark of a loop that cannot see what it optimises.)"""
    if occluding:  # the round-1 layout, kept for like-for-like comparisons with round-1 numbers (bench.py --scene r01)
        return random_gaussians(P, seed=seed, box=0.6, log_scale=log_scale, channels=channels, center=target)
    g = random_gaussians(P, seed=seed, box=1.0, log_scale=log_scale, channels=channels)
    rng = np.random.RandomState(seed + 1000)
    if ring:
        ang = rng.uniform(0, 2 * math.pi, size=P)
        rad = rng.uniform(2.2, 2.6, size=P)
        xyz = np.stack([target[0] + rad * np.sin(ang), rng.uniform(-0.5, 1.1, size=P), target[2] + rad * np.cos(ang)], 1)
    else:
        xyz = np.stack([target[0] + rng.uniform(-1.5, 1.5, size=P), target[1] + rng.uniform(-0.8, 0.8, size=P),
                        target[2] - rng.uniform(0.4, 1.0, size=P)], 1)
    g["means3D"] = xyz.astype(np.float32)
    return g


def smoke_scene(P_fluid, P_background, seed=0, channels=3, ring=False, occluding=False):
    """config 3/4/5 cloud: fluid plume (visual particles) in front of static background Gaussians."""
    fluid = plume_gaussians(P_fluid, seed=seed, channels=channels)
    bgd = backdrop_gaussians(P_background, seed=seed + 1, ring=ring, channels=channels, occluding=occluding)
    return {k: np.concatenate([fluid[k], bgd[k]], axis=0) for k in fluid}


def arc_cameras(n, W, H, target=(0.34, 0.3, -0.225), distance=1.6, arc_deg=120.0, fov=0.8, height=0.3, device="cuda"):
    """n views on a horizontal arc of `arc_deg` degrees around `target` (README: ~120 degree arc)."""
    cams = []
    for i in range(n):
        a = math.radians(-arc_deg / 2 + arc_deg * (i / max(n - 1, 1))) if n > 1 else 0.0
        eye = (target[0] + distance * math.sin(a), height, target[2] + distance * math.cos(a))
        R, T = look_at(eye, target)
        cams.append(Camera(R, T, fov, fov, W, H, uid=i, device=device))
    return cams


def ring_cameras(n, W, H, target=(0.34, 0.3, -0.225), distance=1.6, fov=0.8, height=0.3, device="cuda"):
    """n views evenly spaced on a full 360 degree ring around `target` (BASELINE config 5: 8 synthetic views)."""
    cams = []
    for i in range(n):
        a = 2.0 * math.pi * i / n
        eye = (target[0] + distance * math.sin(a), height, target[2] + distance * math.cos(a))
        R, T = look_at(eye, target)
        cams.append(Camera(R, T, fov, fov, W, H, uid=i, device=device))
    return cams


def front_camera(W, H, distance=2.0, fov=0.8, device="cuda"):
    """config 1 camera: on +z at `distance`, looking at the origin."""
    R, T = look_at((0.0, 0.0, distance), (0.0, 0.0, 0.0))
    return Camera(R, T, fov, fov, W, H, uid=0, device=device)


def lattice_particles(n_side, spacing=0.9, jitter=0.1, seed=0, origin=(0.0, 0.0, 0.0)):
    """Hidden (physics) particles on a jittered lattice in scaled units (x100): with spacing 0.9
    and H = 2.0 every particle has < KNN_K = 100 neighbours."""
    rng = np.random.RandomState(seed)
    g = np.stack(np.meshgrid(*[np.arange(n_side)] * 3, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.float64)
    x = g * spacing + rng.uniform(-jitter, jitter, size=g.shape) + np.asarray(origin)
    return x.astype(np.float32)


def to_torch(d, device="cuda"):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in d.items()}
