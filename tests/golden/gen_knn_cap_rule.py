"""Writes tests/golden/knn_cap_rule.npz: the max_num_neighbors rule this repository adopts for the reference's
`radius(x, y, r, max_num_neighbors=K)` / `radius_graph(x, r, max_num_neighbors=K)` calls (gm_dynamics.py:1276, 1302, 1463),
as DATA -- a point cloud, r, K and the edge list the rule keeps -- so that wherever torch_cluster is importable the
comparison with the real library is one test (tests/test_knn_cap_rule.py).  torch_cluster is neither in this image nor
vendored by the reference: the rule ("per query, the first K hits in index order of x", what the package's CUDA kernel does:
one thread per query walks x in index order and stops at K) is RESTATED, not pinned.

    python tests/golden/gen_knn_cap_rule.py
"""
import os

import numpy as np


def edges_capped(y, x, r, K):
    """(row = query index into y, col = neighbour index into x) of the kept pairs, row-major."""
    d2 = ((y[:, None, :].astype(np.float64) - x[None, :, :].astype(np.float64)) ** 2).sum(-1)
    hit = d2 < float(r) ** 2
    rank = np.cumsum(hit, axis=1)  # 1-based rank of every hit in index order of x
    keep = hit & (rank <= K)
    row, col = np.nonzero(keep)
    return row.astype(np.int64), col.astype(np.int64)


def main():
    rng = np.random.RandomState(20260930)
    g = np.stack(np.meshgrid(np.arange(7), np.arange(9), np.arange(6), indexing="ij"), -1).reshape(-1, 3)
    x = (g + rng.uniform(-0.25, 0.25, size=g.shape)).astype(np.float32)
    x = x[rng.permutation(x.shape[0])]          # index order unrelated to position: "first K by index" is a scattered subset
    y = (rng.uniform(-0.5, 6.5, size=(150, 3)) * np.array([1.0, 8.5 / 6.5, 5.5 / 6.5])).astype(np.float32)
    r = 2.0
    out = dict(x=x, y=y, r=np.float32(r))
    for K in (1, 5, 12, 100):
        row, col = edges_capped(y, x, r, K)          # radius(x, y, r, max_num_neighbors=K) -> (row into y, col into x)
        out[f"radius_K{K}_row"], out[f"radius_K{K}_col"] = row, col
        row, col = edges_capped(x, x, r, K)          # radius_graph(x, r, loop=True, max_num_neighbors=K): self pairs kept
        out[f"graph_K{K}_row"], out[f"graph_K{K}_col"] = row, col  # (the reference's calls pass loop=True, gmd:1081,1276,1302)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "knn_cap_rule.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
