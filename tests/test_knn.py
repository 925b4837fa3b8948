"""simple-knn distCUDA2: CPU tests pin the oracle with closed-form cases; the GPU test compares the
grid-search kernel with the oracle on uniform, clustered and degenerate clouds."""
import numpy as np
import pytest

from oracle.knn_oracle import mean_dist2_3nn


def test_knn_oracle_closed_forms():
    # unit cubic lattice: every interior point has 6 neighbours at distance 1 -> mean of three 1s
    g = np.stack(np.meshgrid(*[np.arange(5)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    d = mean_dist2_3nn(g)
    interior = ((g > 0) & (g < 4)).all(1)
    assert np.allclose(d[interior], 1.0)
    corner = (g == 0).all(1)
    assert np.allclose(d[corner], 1.0)                      # a corner still has 3 axis neighbours
    # collinear points 0, 1, 3, 7: nearest three of 0 are 1, 3, 7 -> (1 + 9 + 49) / 3
    line = np.array([[0, 0, 0], [1, 0, 0], [3, 0, 0], [7, 0, 0]], np.float32)
    assert np.allclose(mean_dist2_3nn(line), [(1 + 9 + 49) / 3, (1 + 4 + 36) / 3, (4 + 9 + 16) / 3, (16 + 36 + 49) / 3])
    # duplicates count as neighbours at distance 0 (only the query's own index is skipped, simple_knn.cu:158)
    dup = np.array([[0, 0, 0], [0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32)
    assert np.allclose(mean_dist2_3nn(dup)[0], (0 + 1 + 4) / 3)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uniform", "clustered", "outliers", "tiny", "plume"])
def test_knn_matches_oracle(kind):
    import torch
    from simple_knn._C import distCUDA2
    rng = np.random.RandomState(5)
    if kind == "uniform":
        p = rng.uniform(-1, 1, size=(12000, 3))
    elif kind == "clustered":
        p = np.concatenate([rng.normal(size=(3000, 3)) * 0.01 + c for c in rng.uniform(-2, 2, size=(5, 3))])
    elif kind == "outliers":
        p = np.concatenate([rng.uniform(0, 1, size=(5000, 3)), rng.uniform(-500, 500, size=(7, 3)), np.zeros((3, 3))])
    elif kind == "tiny":
        p = rng.uniform(0, 1, size=(5, 3))
    else:
        from fluidnexus_amd import synthetic as S
        p = S.plume_gaussians(20000, seed=3)["means3D"] * 100.0
    p = p.astype(np.float32)
    got = distCUDA2(torch.tensor(p).cuda()).cpu().numpy()
    ref = mean_dist2_3nn(p)
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-12, np.abs(got - ref).max()
    rel = np.abs(got - ref) / (ref + 1e-30)
    assert rel.max() <= 1e-5


@pytest.mark.gpu
def test_knn_fewer_than_four_points():
    import torch
    from simple_knn._C import distCUDA2
    p = torch.tensor([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0]]).cuda()
    out = distCUDA2(p).cpu().numpy()
    assert out.shape == (2,) and (out > 1e37).all()       # FLT_MAX terms, as the reference (simple_knn.cu:140,165)
    assert distCUDA2(torch.zeros(0, 3).cuda()).shape == (0,)
