"""CPU restatement of simple-knn's result -- TEST INFRASTRUCTURE ONLY.
FluidDynamics/submodules/simple-knn/simple_knn.cu:134-166 (boxMeanDist) returns, for every point, the mean of
the squared fp32 distances to its 3 nearest OTHER points, found exactly (the Morton order and the box pruning
only decide which candidates are visited, :150-156); updateKBest keeps the three smallest (:120-131) and the
result is (best0 + best1 + best2) / 3.0f (:165).  This restatement evaluates that definition by brute force.

PARITY STATUS: parity unpinned against the CUDA build (cannot be compiled or run here); the definition is
algorithm-independent, and nvcc's FMA contraction of dx*dx + dy*dy + dz*dz is the only unknowable (<= 1 ulp per
distance).  Pinned by closed-form cases (lattices) in tests/test_knn.py."""
import numpy as np


def mean_dist2_3nn(points, chunk=2048):
    p = np.ascontiguousarray(points, dtype=np.float32)
    N = p.shape[0]
    out = np.empty(N, np.float32)
    for a in range(0, N, chunk):
        q = p[a:a + chunk]
        d = q[:, None, :] - p[None, :, :]                                # fp32, point - ref (:121)
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        d2[np.arange(q.shape[0]), np.arange(a, a + q.shape[0])] = np.float32(np.finfo(np.float32).max)  # i == idx skipped
        if N - 1 < 3:
            pad = np.full((q.shape[0], 3), np.finfo(np.float32).max, np.float32)
            d2 = np.concatenate([d2, pad], 1)
        best = np.sort(np.partition(d2, 2, axis=1)[:, :3], axis=1)
        out[a:a + chunk] = ((best[:, 0] + best[:, 1]) + best[:, 2]) / np.float32(3.0)
    return out
