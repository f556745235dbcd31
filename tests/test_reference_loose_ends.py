"""Two places where the reference's behaviour is NOT reproduced bit for bit, bounded here.

1. Sophus.  The reference kernel builds `Sophus::SE3<float> pose(Matrix4f)` (layers/sdf_matching_loss_kernel.cu:125-126):
   the rotation goes through a unit quaternion (Eigen's matrix -> quaternion conversion), points are rotated with the
   quaternion formula and gradients are rotated back with the matrix regenerated from the quaternion.  The oracle and the
   HIP kernels multiply by the float32 matrix directly.  Both are float32 evaluations of the same rigid motion; the test
   restates Sophus' path in float32 numpy and bounds the difference in object-space coordinates, potentials and gradients
   for float32-rounded inverse poses as omg.util.se3_inverse produces them.
2. Ties in the top-k set.  The reference takes `np.argsort(potentials.flatten())[-k:]` (omg/cost.py:392) with numpy's
   default introsort, which is not stable: among EQUAL potentials at the cut, which indices make it into the set — and which
   of several equal maxima of a (waypoint, link) group is "written last" — depends on the numpy build.  Oracle and HIP define
   it: ties rank by ascending flat index (a stable sort).  The test builds a scene with exact ties (box SDFs have planes of
   equal values) and checks what can and cannot differ.
"""
from __future__ import annotations

import numpy as np

from omg_planner_amd import robot as rb, scenes as sc
from oracle import oracle as orc

F = np.float32


def _eigen_quaternion_from_matrix(m):
    """Eigen::Quaternion<float>(Matrix3f) (Eigen/src/Geometry/Quaternion.h, quaternionbase_assign_impl), float32."""
    m = m.astype(F)
    t = F(m[0, 0] + m[1, 1] + m[2, 2])
    q = np.zeros(4, F)  # x, y, z, w
    if t > 0:
        t = np.sqrt(F(t + F(1)))
        q[3] = F(0.5) * t
        t = F(0.5) / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(F(m[i, i] - m[j, j] - m[k, k] + F(1)))
        q[i] = F(0.5) * t
        t = F(0.5) / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q


def _sophus_rotate(q, p):
    """Sophus::SO3::operator*(Point): uv = 2 q.vec x p;  p + w uv + q.vec x uv  (float32)."""
    v, w = q[:3].astype(F), F(q[3])
    uv = np.cross(v, p).astype(F)
    uv = (uv + uv).astype(F)
    return (p + w * uv + np.cross(v, uv).astype(F)).astype(F)


def _quaternion_matrix(q):
    """Eigen::Quaternion::toRotationMatrix (what pose.rotationMatrix() returns), float32."""
    x, y, z, w = (F(v) for v in q)
    tx, ty, tz = F(2) * x, F(2) * y, F(2) * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[F(1) - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, F(1) - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, F(1) - (txx + tyy)]], F)


def test_sophus_quaternion_round_trip_is_below_the_parity_bar():
    rng = np.random.RandomState(11)
    worst_u = worst_pot = worst_grad = 0.0
    flips = total = 0
    for trial in range(12):
        # a random rigid pose, inverted and rounded to float32 exactly as the host side does (omg/util.py:129-135)
        A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        A *= np.sign(np.linalg.det(A))
        pose = np.eye(4)
        pose[:3, :3], pose[:3, 3] = A, rng.uniform(-0.8, 0.8, 3)
        inv = sc.se3_inverse(pose).astype(F)
        grid = sc.sphere_sdf(0.08, (24, 24, 24), 0.5 / 24) if trial % 2 else sc.box_sdf((0.06, 0.09, 0.05), (24, 20, 28), 0.02)
        sdf, lim = sc.pack_padded([sc.SceneObject("o", pose, grid)])
        # world points whose object-space image lies inside / around the grid
        local = rng.uniform(grid.min_coords - 0.03, grid.min_coords + np.array(grid.data.shape) * grid.delta + 0.03, (6000, 3))
        pts = (local @ A.T + pose[:3, 3]).astype(F)
        eps, pad, clr, dis = (np.array([v], F) for v in (0.2, 1.0, 0.01, 0.0))
        pot_a, grad_a, col_a = orc.sdf_loss_forward(inv[None], sdf, lim, pts, eps, pad, clr, dis)
        # the Sophus path: quaternion rotation + translation in float32, then the SAME lookup with the identity pose
        q = _eigen_quaternion_from_matrix(inv[:3, :3])
        qn = np.sqrt(float((q.astype(np.float64) ** 2).sum()))
        assert abs(qn - 1.0) < 2e-6  # the conversion of an orthonormal float32 matrix is a unit quaternion up to rounding
        u_b = np.stack([_sophus_rotate(q, p) + inv[:3, 3] for p in pts]).astype(F)
        u_a = (pts.astype(np.float64) @ inv[:3, :3].astype(np.float64).T + inv[:3, 3]).astype(F)
        worst_u = max(worst_u, float(np.abs(u_a - u_b).max()))
        pot_b, grad_o, col_b = orc.sdf_loss_forward(np.eye(4, dtype=F)[None], sdf, lim, u_b, eps, pad, clr, dis)
        grad_b = (grad_o.astype(F) @ _quaternion_matrix(q)).astype(F)  # rotationMatrix().transpose() * vgrad (.cu:176-179)
        worst_pot = max(worst_pot, float(np.abs(pot_a - pot_b).max()))
        worst_grad = max(worst_grad, float(np.abs(grad_a - grad_b).max()))
        flips += int((col_a != col_b).sum())
        total += len(pts)
        assert (pot_a != 0).mean() > 0.05
    # object-space coordinates agree to a few float32 ulps of a metre-sized coordinate; potentials follow (|d pot / d u| <= 1);
    # gradients are central differences over one voxel (2 cm), so a coordinate change of 2e-7 m moves them by ~1e-5 at most
    assert worst_u < 1e-6, worst_u
    assert worst_pot < 1e-6, worst_pot
    assert worst_grad < 5e-5, worst_grad
    assert flips <= 2e-4 * total, (flips, total)  # `collides` is a threshold: a value within 1e-7 of the clearance may flip


def test_top_k_ties_differ_from_numpy_only_inside_the_tie_at_the_cut():
    """Exact ties at the top-k cut on a box-shaped SDF without the fixtures' 1e-4 ramp."""
    m = rb.PandaModel(seed=0)
    P, blob = m.points_per_link, m.blob()
    # a big flat box right under the arm's sweep: the potentials of all points at the same height inside the band are equal
    box = sc.box_sdf((0.6, 0.6, 0.02), (48, 48, 16), 0.03)
    scene = sc.Scene([sc.SceneObject("table", sc._yaw_pose(0.45, 0.0, 0.25, 0.0), box)], 0)
    batch = sc.pack_table([scene], dict(epsilon=0.2, target_epsilon=0.2, clearance=0.01, target_clearance=0.01))
    goal = np.array([0.3, 0.2, 0.1, -1.6, 0.1, 1.9, 1.0, 0.04, 0.04])
    traj = sc.linear_init(rb.HOME_CONFIG, goal, 30)[None]
    traj[0, :, 7:] = 0.04  # both fingers at the same opening: mirrored finger points sit at equal heights
    pot, _, _ = orc.fk_sdf(blob, P, batch, traj)
    flat = pot[0].ravel()
    nz = flat[flat > 0]
    vals, counts = np.unique(nz, return_counts=True)
    assert (counts > 1).sum() > 0, "the scene must produce exactly equal non-zero potentials"
    found = False
    for k in range(10, len(nz)):
        cut = np.sort(flat)[-k]
        ours = set(np.argsort(flat, kind="stable")[-k:].tolist())        # oracle / HIP: ties rank by ascending flat index
        theirs = set(np.argsort(flat)[-k:].tolist())                      # the reference: numpy's default (unstable) sort
        diff = ours ^ theirs
        # whatever numpy does, the two sets can only differ in elements EQUAL to the cut value, and hold equally many of them
        assert all(flat[i] == cut for i in diff)
        assert len(ours - theirs) == len(theirs - ours)
        np.testing.assert_array_equal(np.sort(flat[list(ours)]), np.sort(flat[list(theirs)]))  # same multiset of potentials
        n_cut = int((flat == cut).sum())
        taken = int(sum(flat[i] == cut for i in ours))
        if n_cut > taken > 0:  # the cut falls INSIDE a tie group: this is where the builds may pick different points
            found = True
            # ours: the highest flat indices of the group
            idx = np.flatnonzero(flat == cut)
            assert sorted(i for i in ours if flat[i] == cut) == idx[-taken:].tolist()
    assert found, "no k puts the cut inside a tie group: the test scene lost its ties"


def test_oracle_sophus_mode_is_the_numpy_restatement():
    """The oracle's second mode (oracle/omg_oracle.c, SOPHUS MODE: quaternion from the float32 matrix, quaternion rotation, gradient
    rotated back with the regenerated matrix) against the float32 numpy restatement above: same coordinates, hence the same
    potentials and collision flags bit for bit; gradients up to the association of a three-term float32 sum."""
    rng = np.random.RandomState(5)
    assert not orc.sophus_mode()
    for trial in range(6):
        A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        A *= np.sign(np.linalg.det(A))
        if trial == 4:  # a rotation by ~pi: the trace is negative, Eigen's conversion takes its other branch
            A = np.diag([1.0, -1.0, -1.0]) @ np.linalg.qr(np.eye(3) + 0.01 * rng.normal(size=(3, 3)))[0]
            A *= np.sign(np.linalg.det(A))
        pose = np.eye(4)
        pose[:3, :3], pose[:3, 3] = A, rng.uniform(-0.8, 0.8, 3)
        inv = sc.se3_inverse(pose).astype(F)
        grid = sc.sphere_sdf(0.08, (24, 24, 24), 0.5 / 24) if trial % 2 else sc.box_sdf((0.06, 0.09, 0.05), (24, 20, 28), 0.02)
        sdf, lim = sc.pack_padded([sc.SceneObject("o", pose, grid)])
        local = rng.uniform(grid.min_coords - 0.03, grid.min_coords + np.array(grid.data.shape) * grid.delta + 0.03, (3000, 3))
        pts = (local @ A.T + pose[:3, 3]).astype(F)
        eps, pad, clr, dis = (np.array([v], F) for v in (0.2, 1.0, 0.01, 0.0))
        q = _eigen_quaternion_from_matrix(inv[:3, :3])
        if trial == 4:
            assert float(inv[0, 0] + inv[1, 1] + inv[2, 2]) <= 0
        u_b = np.stack([_sophus_rotate(q, p) + inv[:3, 3] for p in pts]).astype(F)
        pot_b, grad_o, col_b = orc.sdf_loss_forward(np.eye(4, dtype=F)[None], sdf, lim, u_b, eps, pad, clr, dis)
        grad_b = (grad_o.astype(np.float64) @ _quaternion_matrix(q).astype(np.float64))
        try:
            orc.set_sophus_mode(True)
            pot_s, grad_s, col_s = orc.sdf_loss_forward(inv[None], sdf, lim, pts, eps, pad, clr, dis)
        finally:
            orc.set_sophus_mode(False)
        assert (pot_s != 0).mean() > 0.05
        np.testing.assert_array_equal(pot_s, pot_b)
        np.testing.assert_array_equal(col_s, col_b)
        np.testing.assert_allclose(grad_s, grad_b, rtol=0, atol=2e-6)
    assert not orc.sophus_mode()
