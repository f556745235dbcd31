"""The far box of an object record may be any box outside which a lookup adds nothing.  scenes.tighten_far_boxes
shrinks it to the voxels that can matter; here the oracle confirms that points outside add exactly nothing."""
import numpy as np
import pytest

from omg_planner_amd import scenes as sc
from oracle import oracle as orc


def _one_object_check(grid, delta, eps, clr, rng, n=60000):
    dims = np.array(grid.shape)
    lo = np.array([-0.3, 0.1, -0.05])
    ob = sc.SceneObject("o", np.eye(4), sc.SdfGrid(grid, lo, delta))
    rec = np.zeros(1, sc.OBJECT_DTYPE)
    rec["pose_inv"] = np.eye(4, dtype=np.float32)[:3].ravel()
    rec["lo"] = ob.sdf.min_coords
    rec["hi"] = ob.sdf.max_coords
    rec["dim"] = dims
    rec["delta"] = delta
    rec["epsilon"], rec["padding_scale"], rec["clearance"] = eps, 1.0, clr
    sc.finish_records(rec)
    loose = rec.copy()
    sc.tighten_far_boxes(rec, grid.ravel())
    r = rec[0]
    assert (r["far_lo"] >= loose[0]["far_lo"]).all() and (r["far_hi"] <= loose[0]["far_hi"]).all()
    # points all over the (loose) box, with extra density near the faces of the tight one
    ext = (r["hi"] - r["lo"]).astype(np.float64)
    t = rng.uniform(-2 * delta, ext + 2 * delta, size=(n, 3))
    for a in range(3):
        for edge in (r["far_lo"][a], r["far_hi"][a]):
            if np.isfinite(edge):
                k = rng.randint(0, n, n // 12)
                t[k, a] = edge + rng.uniform(-1.5, 1.5, len(k)) * delta
    pts = (t + r["lo"]).astype(np.float32)
    tt = pts - r["lo"]  # the kernel's float32 offset (identity pose)
    outside = ~np.all((tt >= r["far_lo"]) & (tt <= r["far_hi"]), axis=1)
    lim = np.concatenate([r["lo"], r["hi"], dims.astype(np.float32), [np.float32(delta)]]).astype(np.float32)[None]
    pot, grad, col = orc.sdf_loss_forward(np.eye(4, dtype=np.float32)[None], grid[None], lim, pts, np.float32([eps]),
                                          np.float32([1.0]), np.float32([clr]), np.float32([0.0]))
    assert outside.any()
    assert not pot[outside].any() and not grad[outside].any() and not col[outside].any()
    return outside.mean(), (pot != 0).mean()


@pytest.mark.parametrize("eps,clr", [(0.1, 0.0), (0.2, 0.01), (0.05, 0.0)])
def test_tight_box_drops_only_silent_points(eps, clr):
    rng = np.random.RandomState(3)
    g = sc.sphere_sdf(0.07, shape=(40, 40, 40), delta=0.6 / 40)
    frac_out, frac_pot = _one_object_check(g.data, g.delta, eps, clr, rng)
    assert frac_out > 0.3 and frac_pot > 0.01


def test_tight_box_off_centre_and_extrapolated_edge():
    """Surface close to the low faces: base index 0 extrapolates with negative weights (.cu:39-48)."""
    rng = np.random.RandomState(4)
    delta = 0.02
    x, y, z = np.meshgrid(*[(np.arange(d) + 0.5) * delta for d in (24, 30, 20)], indexing="ij")
    grid = (np.sqrt((x - 0.03) ** 2 + (y - 0.5) ** 2 + (z - 0.2) ** 2) - 0.05).astype(np.float32)
    _one_object_check(grid, delta, 0.08, 0.01, rng)
    # noisy field: nothing Lipschitz about it
    grid2 = (grid + rng.normal(0, 0.05, grid.shape)).astype(np.float32)
    _one_object_check(grid2, delta, 0.08, 0.01, rng)


def test_no_reachable_voxel_gives_an_empty_box():
    grid = np.full((8, 9, 10), 0.9, np.float32)
    assert sc.influence_range(grid, 0.2, 0.01) is None
    rec = np.zeros(1, sc.OBJECT_DTYPE)
    rec["hi"] = 1.0
    rec["dim"] = grid.shape
    rec["delta"] = 0.1
    rec["epsilon"], rec["clearance"] = 0.2, 0.01
    sc.finish_records(rec)
    sc.tighten_far_boxes(rec, grid.ravel())
    assert (rec["far_lo"] > rec["far_hi"]).all()


def test_uncullable_records_keep_the_grid_box():
    grid = np.full((8, 9, 10), 0.9, np.float32)
    rec = np.zeros(1, sc.OBJECT_DTYPE)
    rec["hi"] = 1.0
    rec["dim"] = grid.shape
    rec["delta"] = 0.1
    rec["epsilon"], rec["clearance"] = 1.5, 0.01   # value 1.0 outside the grid is inside the hinge
    sc.finish_records(rec)
    before = rec.copy()
    sc.tighten_far_boxes(rec, grid.ravel())
    assert np.array_equal(before["far_lo"], rec["far_lo"]) and np.array_equal(before["far_hi"], rec["far_hi"])
