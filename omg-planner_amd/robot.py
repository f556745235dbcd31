"""Panda robot model for the CHOMP engine: kinematic constants, joint limits, collision points.

Host-side mirror of what the hot path reads from the reference's ``Robot`` / ``robot_kinematics``
objects (omg/core.py:143-190, ycb_render/robotPose/robot_pykdl.py:96-113):

* ``robot.robot_kinematics._pose_0 / _tip2joint / _joint_axis / center_offset``  -> ``PandaModel`` tables
* ``robot.collision_points [10, P, 3]``                                          -> ``collision_points``
* ``robot.joint_lower_limit / joint_upper_limit [1, 9]`` (URDF limits -/+ 0.2)    -> ``joint_lower_limit`` ...

``blob()`` packs them into the flat float64 layout of ``include/omg_hip.h`` (OMGX_ROBOT_*).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

NUM_LINKS = 10
NUM_DOF = 9
_DATA = Path(__file__).resolve().parent / "data" / "panda_fk.npz"

# default start configuration of Trajectory (omg/core.py:38)
HOME_CONFIG = np.array([0.0, -1.285, 0.0, -2.356, 0.0, 1.571, 0.785, 0.04, 0.04])


class PandaModel:
    """Kinematic constants + sampled collision points of the Franka Panda (7 arm dof + 2 fingers)."""

    def __init__(self, collision_points: np.ndarray | None = None, soft_joint_limit_padding: float = 0.2,
                 points_per_link: int = 15, seed: int = 0):
        d = np.load(_DATA)
        self.pose_0 = np.ascontiguousarray(d["pose_0"], dtype=np.float64)
        self.tip2joint = np.ascontiguousarray(d["tip2joint"], dtype=np.float64)
        self.center_offset = np.ascontiguousarray(d["center_offset"], dtype=np.float64)
        self.joint_axis = np.ascontiguousarray(d["joint_axis"], dtype=np.float64)
        limits = np.asarray(d["joint_limits"], dtype=np.float64)
        # omg/core.py:157-164: arm joints padded inwards, fingers untouched
        self.joint_lower_limit = limits[:, 0][None].copy()
        self.joint_upper_limit = limits[:, 1][None].copy()
        self.joint_lower_limit[:, :-2] += soft_joint_limit_padding
        self.joint_upper_limit[:, :-2] -= soft_joint_limit_padding
        if collision_points is None:
            collision_points = synthetic_collision_points(points_per_link, seed)
        self.collision_points = np.ascontiguousarray(collision_points, dtype=np.float64)
        assert self.collision_points.shape[0] == NUM_LINKS and self.collision_points.shape[2] == 3

    @property
    def points_per_link(self) -> int:
        return int(self.collision_points.shape[1])

    def blob(self) -> np.ndarray:
        """Flat float64 constants blob: raw tables (include/omg_hip.h OMGX_ROBOT_*) followed by the
        derived constants the device FK uses (same header, "followed, at D = 528+30P, by ...")."""
        raw = np.concatenate([
            self.pose_0.ravel(), self.tip2joint.ravel(), self.center_offset.ravel(), self.joint_axis.ravel(),
            self.joint_lower_limit.ravel(), self.joint_upper_limit.ravel(), self.collision_points.ravel(),
        ]).astype(np.float64)
        return np.concatenate([raw, self._derived()])

    def _derived(self) -> np.ndarray:
        """Re-association of forward_kinematics_parallel (robot_pykdl.py:148-215) into 3x3 pieces.

        b_i(q) = pose_0[i] @ Rz(q) @ Rx(off_i), columns 1,2 negated for i > 0 (line 176).  With
        Rz(q) = cos(q) E1 + sin(q) E2 + E3 its rotation part is cos(q) U_i + sin(q) V_i + W_i and its
        translation is pose_0[i][:3,3].  center_offset is folded into the collision points and
        tip2joint into the joint axis / origin, so the device never forms a 4x4 product.
        """
        offs = [0.0, -np.pi, np.pi, np.pi, -np.pi, np.pi, np.pi]
        E1 = np.diag([1.0, 1.0, 0.0])
        E2 = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
        E3 = np.diag([0.0, 0.0, 1.0])
        uvw, tp = [], []
        for i in range(7):
            co, so = np.cos(offs[i]), np.sin(offs[i])
            Rx = np.array([[1.0, 0.0, 0.0], [0.0, co, -so], [0.0, so, co]])
            N = np.eye(3) if i == 0 else np.diag([1.0, -1.0, -1.0])
            C = Rx @ N
            Rp = self.pose_0[i][:3, :3]
            uvw.append(np.stack([Rp @ E1 @ C, Rp @ E2 @ C, Rp @ E3 @ C]))
            tp.append(self.pose_0[i][:3, 3])
        rows = lambda T: T[:3, :4].ravel()
        P = self.points_per_link
        pts = np.einsum("lrc,lpc->lpr", self.center_offset[:, :3, :3], self.collision_points) + self.center_offset[:, None, :3, 3]
        ax = np.einsum("lrc,lc->lr", self.tip2joint[:, :3, :3], self.joint_axis)
        og = self.tip2joint[:, :3, 3]
        radius = np.linalg.norm(pts, axis=-1).max(axis=1)  # bounding-sphere radius of each link's centred points
        key = pts.tobytes()
        if key not in _BALL_CACHE:
            if len(_BALL_CACHE) > 64:
                _BALL_CACHE.clear()
            _BALL_CACHE[key] = np.stack([bounding_ball(pts[l]) for l in range(NUM_LINKS)])
        ball = _BALL_CACHE[key]  # [10][4]: a small ball around the points themselves
        out = np.concatenate([np.array(uvw).ravel(), np.array(tp).ravel(), rows(self.pose_0[7]), rows(self.pose_0[8]),
                              rows(self.pose_0[9]), pts.ravel(), ax.ravel(), og.ravel(), radius.ravel(), ball.ravel()])
        assert out.size == 356 + 30 * P
        return out


_BALL_CACHE: dict = {}


def bounding_ball(points: np.ndarray) -> np.ndarray:
    """(cx, cy, cz, r): the smallest ball that holds every point (Welzl's algorithm on the at most 16 points of a link:
    deterministic — the points are taken in their given order), r rounded up.  The row-level culling tests this ball instead of
    the one about the link's frame origin (RAD): with the points 2-10 cm off the origin it is about half as large, and a third
    of the main loop's far tests never start.  Any centre is valid: r is recomputed as the largest
    distance to it."""
    p = np.asarray(points, np.float64)

    def ball_of(sup):
        if len(sup) == 0:
            return np.zeros(3), -1.0
        if len(sup) == 1:
            return sup[0].copy(), 0.0
        q0 = sup[0]
        A = np.array([q - q0 for q in sup[1:]])            # the centre lies in q0 + span(A) and is equidistant from the support
        b = 0.5 * (A * A).sum(axis=1)
        lam = np.linalg.lstsq(A @ A.T, b, rcond=None)[0]
        c = q0 + A.T @ lam
        return c, float(np.sqrt(((sup[0] - c) ** 2).sum()))

    def inside(q, c, r):
        return r >= 0.0 and ((q - c) ** 2).sum() <= r * r * (1.0 + 1e-12) + 1e-30

    def welzl(n, sup):
        c, r = ball_of(sup)
        if len(sup) == 4:
            return c, r
        for i in range(n):
            if not inside(p[i], c, r):
                c, r = welzl(i, sup + [p[i]])
        return c, r

    c, _ = welzl(len(p), [])
    r = float(np.sqrt(((p - c) ** 2).sum(axis=1).max())) * (1.0 + 1e-12) + 1e-15
    return np.array([c[0], c[1], c[2], r])


def synthetic_collision_points(points_per_link: int = 15, seed: int = 0) -> np.ndarray:
    """Stand-in for Robot.load_collision_points (omg/core.py:166-190).

    The reference samples ``points_per_link`` surface points per link from ``data/robots/*.xyz`` with
    an unseeded ``random.sample``; that data set is not redistributable/available, so tests and the
    benchmark use points uniform in a 10 cm cube around each (centred) link frame.
    """
    rng = np.random.RandomState(seed)
    return rng.uniform(-0.05, 0.05, size=(NUM_LINKS, points_per_link, 3))
