"""CPU tests: known-answer checks of the oracle's SDF op (layers/sdf_matching_loss_kernel.cu:15-181).

The reference has no CPU implementation and no test vectors for this op ("parity unpinned"), so the
restatement is checked against closed forms: trilinear interpolation reproduces a field that is
linear in the grid coordinates exactly, which fixes value, central-difference gradient, both hinge
branches, the rotate-back, the collides count, the -0.5 voxel-centre shift, truncation toward zero
and the out-of-range 1.0."""
import numpy as np
import pytest

from oracle import oracle as orc


def _linear_grid(shape, a, b):
    """value at voxel index (i,j,k) = a . (i,j,k) + b ; lookup at grid coord g returns a.(g-0.5)+b."""
    I, J, K = np.meshgrid(*[np.arange(s) for s in shape], indexing="ij")
    return (a[0] * I + a[1] * J + a[2] * K + b).astype(np.float32)


def _scene(grid, lo, hi, delta, R=np.eye(3), t=np.zeros(3), eps=0.2, pad=1.0, clr=0.01, dis=0.0):
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = R
    T[:3, 3] = t
    lim = np.array([[*lo, *hi, *grid.shape, delta]], np.float32)
    return T[None], grid[None], lim, np.array([eps], np.float32), np.array([pad], np.float32), \
        np.array([clr], np.float32), np.array([dis], np.float32)


def test_linear_field_value_gradient_and_hinges():
    shape = (12, 10, 8)
    delta = 0.05
    lo = np.array([-0.3, -0.25, -0.2])
    hi = lo + delta * np.array(shape)
    a = np.array([0.05, -0.02, 0.04])  # per-voxel slopes
    grid = _linear_grid(shape, a, -0.25)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    t = np.array([0.01, -0.02, 0.03])
    rng = np.random.RandomState(0)
    q = lo + (hi - lo) * rng.uniform(0.2, 0.8, size=(200, 3))  # object-frame points well inside
    p = (q - t) @ R  # world points with R p + t = q
    pose, g, lim, eps, pad, clr, dis = _scene(grid, lo, hi, delta, R, t)
    pot, grad, col = orc.sdf_loss_forward(pose, g, lim, p.astype(np.float32), eps, pad, clr, dis)
    gc = (q - lo) / (hi - lo) * np.array(shape)
    value = (gc - 0.5) @ a - 0.25
    gobj = a / delta  # central difference one voxel apart, divided by delta
    exp_pot = np.where(value <= 0, -value + 0.1, np.where(value <= 0.2, (value - 0.2) ** 2 / 0.4, 0.0))
    vg = np.where((value <= 0)[:, None], -gobj[None], np.where((value <= 0.2)[:, None], gobj[None] * ((value - 0.2) / 0.2)[:, None], 0.0))
    exp_grad = vg @ R  # R^T vg as row vectors
    np.testing.assert_allclose(pot, exp_pot, rtol=0, atol=2e-6)
    np.testing.assert_allclose(grad, exp_grad, rtol=0, atol=2e-5)
    np.testing.assert_array_equal(col, (value < 0.01).astype(np.float32))
    assert (value <= 0).any() and ((value > 0) & (value <= 0.2)).any() and (value > 0.2).any()


def test_padding_scale_and_disable():
    shape = (8, 8, 8)
    grid = np.full(shape, 0.1, np.float32)
    lo, hi = np.zeros(3), np.ones(3) * 0.8
    p = np.array([[0.4, 0.4, 0.4]], np.float32)
    pose, g, lim, eps, pad, clr, dis = _scene(grid, lo, hi, 0.1, pad=0.5)
    pot, grad, col = orc.sdf_loss_forward(pose, g, lim, p, eps, pad, clr, dis)
    np.testing.assert_allclose(pot, [0.5 * (0.1 - 0.2) ** 2 / 0.4], atol=1e-7)
    np.testing.assert_array_equal(grad, np.zeros((1, 3), np.float32))
    dis[:] = 1
    pot, grad, col = orc.sdf_loss_forward(pose, g, lim, p, eps, pad, clr, dis)
    assert pot[0] == 0 and col[0] == 0


def test_voxel_centre_shift_truncation_and_out_of_range():
    """pGrid-0.5 in (-1,0) truncates to 0 and EXTRAPOLATES (negative weight); below -1 it is out of range
    and the lookup returns 1.0 (.cu:39-50)."""
    shape = (6, 6, 6)
    a = np.array([0.01, 0.0, 0.0])
    grid = _linear_grid(shape, a, 0.05)
    lo, hi = np.zeros(3), np.ones(3) * 0.6
    delta = 0.1
    pose, g, lim, eps, pad, clr, dis = _scene(grid, lo, hi, delta, eps=0.2, clr=0.0)
    # grid coord gx = 10 * x
    pts = np.array([[0.02, 0.3, 0.3],    # gx=0.2 -> x0=0, fx=-0.3: extrapolated value 0.05-0.003
                    [-0.06, 0.3, 0.3],   # gx=-0.6 -> gx-0.5=-1.1 -> x0=-1: out of range -> 1.0 -> no potential
                    [0.549, 0.3, 0.3],   # gx=5.49 -> x0=4, x1=5 in range
                    [0.551, 0.3, 0.3]],  # gx=5.51 -> x0=5, x1=6 out of range
                   np.float32)
    pot, grad, col = orc.sdf_loss_forward(pose, g, lim, pts, eps, pad, clr, dis)
    v0 = 0.05 + 0.01 * (0.2 - 0.5)
    assert pot[0] == pytest.approx((v0 - 0.2) ** 2 / 0.4, abs=1e-6)
    assert pot[1] == 0 and pot[3] == 0
    v2 = 0.05 + 0.01 * (5.49 - 0.5)
    assert pot[2] == pytest.approx((v2 - 0.2) ** 2 / 0.4, abs=1e-6)
    # point 0: the -x neighbour lookup (gx-1) is out of range -> 1.0 enters the central difference
    g_expected = 0.5 * ((0.05 + 0.01 * (1.2 - 0.5)) - 1.0) / delta
    assert grad[0, 0] == pytest.approx(g_expected * (v0 - 0.2) / 0.2, rel=1e-5)


def test_objects_are_summed_in_index_order_and_collides_counts_objects():
    shape = (8, 8, 8)
    lo, hi = np.zeros(3), np.ones(3) * 0.8
    grids = np.stack([np.full(shape, -0.01, np.float32), np.full(shape, 0.005, np.float32), np.full(shape, 0.5, np.float32)])
    T = np.tile(np.eye(4, dtype=np.float32), (3, 1, 1))
    lim = np.tile(np.array([[*lo, *hi, *shape, 0.1]], np.float32), (3, 1))
    eps = np.array([0.2, 0.1, 0.2], np.float32)
    pad = np.ones(3, np.float32)
    clr = np.array([0.01, 0.0, 0.01], np.float32)
    dis = np.zeros(3, np.float32)
    p = np.array([[0.4, 0.4, 0.4]], np.float32)
    pot, grad, col = orc.sdf_loss_forward(T, grids, lim, p, eps, pad, clr, dis)
    e0 = np.float32(0.01) + np.float32(0.1)
    assert col[0] == 1.0  # object 0 (value < 0.01); object 1 has 0.005 >= clearance 0.0
    assert pot[0] == pytest.approx(float(e0) + (0.005 - 0.1) ** 2 / 0.2, abs=1e-6)
