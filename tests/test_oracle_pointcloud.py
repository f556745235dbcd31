"""CPU test: the oracle's point-cloud SDF (brute force) against scipy's cKDTree query used the way
PointEnv.compute_sdf_from_points uses it (omg/core.py:426-457: np.arange grid axes, "ij" meshgrid, k=1 Euclidean)."""
import numpy as np
from scipy.spatial import cKDTree

from oracle import oracle as orc


def reference_grid(points, res=0.02, margin=0.24):
    bounds = np.stack((points.min(0), points.max(0)), axis=1)
    ax = [np.arange(bounds[a][0] - margin, bounds[a][1] + margin, res) for a in range(3)]
    g = np.array(np.meshgrid(*ax, indexing="ij"))
    d, _ = cKDTree(points).query(g.reshape((3, -1)).T)
    return d.reshape(g.shape[1:]).astype(np.float32), bounds[:, 0] - margin, np.array(g.shape[1:], np.int32)


def test_point_cloud_sdf_matches_ckdtree():
    rng = np.random.RandomState(0)
    for n, box in [(4096, [0.4, 0.6, 0.4]), (17, [0.1, 0.05, 0.2]), (2, [0.0, 0.0, 0.0])]:
        pts = rng.uniform(-0.5, 0.5, size=(n, 3)) * np.array(box) + np.array([0.5, 0.0, 0.2])
        ref, origin, dims = reference_grid(pts)
        got = orc.point_cloud_sdf(pts, origin, 0.02, dims)
        assert got.shape == ref.shape
        np.testing.assert_array_equal(got.view(np.int32), ref.view(np.int32))  # bit for bit (see the test below)


def test_grid_nodes_follow_numpys_arange_fill():
    """np.arange(start, stop, step) yields start, start + step, start + i * ((start + step) - start) — not start + i * step
    (numpy's DOUBLE_fill).  With the naive nodes a few voxels per grid round differently (found by tests/fuzz/fuzz_misc.py)."""
    rng = np.random.RandomState(0)
    naive_differs = 0
    for _ in range(25):
        n = int(rng.choice([1, 2, 17, 300]))
        pts = rng.normal(0, rng.uniform(0.01, 0.3), (n, 3)) + rng.uniform(-1, 1, 3)
        res, margin = float(rng.choice([0.02, 0.05, 0.013])), float(rng.choice([0.24, 0.05, 0.1]))
        ref, origin, dims = reference_grid(pts, res, margin)
        got = orc.point_cloud_sdf(pts, origin, res, dims)
        np.testing.assert_array_equal(got.view(np.int32), ref.view(np.int32))
        ax = np.arange(origin[0], pts[:, 0].max() + margin, res)
        naive_differs += int((origin[0] + np.arange(len(ax)) * res != ax).sum())
    assert naive_differs > 0
