"""ctypes front-end of oracle/omg_oracle.c.  TEST INFRASTRUCTURE ONLY (see the C file's header).

Every function takes/returns numpy arrays; semantics and reference citations are in omg_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libomg_oracle.so"

NUM_LINKS, NUM_DOF, INFO_STRIDE = 10, 9, 16
INFO_KEYS = ["cost", "obs", "smooth", "weighted_obs", "weighted_smooth", "weighted_obs_grad",
             "weighted_smooth_grad", "grad", "collide", "reach", "terminate", "failure_terminate",
             "execute", "standoff_idx", "violate_limit", "limit_steps"]


class ChompParams(C.Structure):
    """Mirror of `omgx_chomp_params` (include/omg_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "n_waypoints", "n_points", "top_k", "consider_finger", "goal_set_proj", "constraint_num",
        "use_standoff", "uncheck_finger_collision", "joint_limit_max_steps", "allow_collision_point",
        "pre_terminate", "do_update")] + [(n, C.c_double) for n in (
        "time_interval", "obstacle_weight", "smoothness_weight", "step_size", "clip_grad_scale",
        "terminate_smooth_loss")] + [("link_smooth_weight", C.c_double * NUM_DOF)] + [(n, C.c_void_p) for n in (
        "waypoint_poses", "start_poses", "end_poses")]


class LearnerParams(C.Structure):
    """Mirror of `omgx_learner_params` (include/omg_hip.h)."""
    _fields_ = [(n, C.c_int32) for n in ("alg", "num_goals", "n_waypoints", "start_idx", "constraint_num", "use_standoff",
                                          "normalize_cost", "cost_parts")] + [(n, C.c_double) for n in (
        "base_obstacle_weight", "smooth_weight", "eta")] + [(n, C.c_void_p) for n in ("goal_pose_table", "end_poses_out")]


ALG = {"FTL": 0, "FTC": 1, "Exp": 2, "MD": 3, "Proj": 4}


def learner_state_init(S: int, G: int) -> np.ndarray:
    """sum_costs 0 | p 1/G | experts_p 1/G | q 1/5 | experts_costs 0  (Learner.__init__, online_learner.py:66-95)."""
    st = np.zeros((S, 7 * G + 10))
    st[:, G:7 * G] = 1.0 / G
    st[:, 7 * G:7 * G + 5] = 0.2
    return st


def build(force: bool = False) -> Path:
    src = _HERE / "omg_oracle.c"
    hdr = _HERE.parent / "include" / "omg_hip.h"
    if force or not _SO.exists() or (src.exists() and _SO.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime)):
        subprocess.run(["make", "-C", str(_HERE), "-B" if force else "-s"], check=True,
                       stdout=subprocess.DEVNULL if not force else None)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_SO))
        for name in ("orc_sdf_loss_forward", "orc_fk_sdf", "orc_goalset_cost", "orc_chomp_optimize", "orc_goal_update", "orc_point_cloud_sdf",
                     "orc_sizeof_object", "orc_sizeof_params", "orc_sizeof_learner_params"):
            getattr(_lib, name).restype = C.c_int
        for name in ("orc_fk_batch", "orc_smooth_matrices", "orc_points_of_config"):
            getattr(_lib, name).restype = None
        assert _lib.orc_sizeof_object() == 184
        assert _lib.orc_sizeof_params() == C.sizeof(ChompParams)
        assert _lib.orc_sizeof_learner_params() == C.sizeof(LearnerParams)
    return _lib


def set_threads(n: int) -> None:
    """OpenMP threads used by the batched oracle loops (cpu_baseline states this number)."""
    omp = C.CDLL("libgomp.so.1")
    omp.omp_set_num_threads(int(n))


def set_sophus_mode(on: bool) -> None:
    """SDF pairs through Sophus' quaternion round trip (layers/sdf_matching_loss_kernel.cu:125-133,176) instead of the float32
    matrix product — see omg_oracle.c (SOPHUS MODE).  A process-wide switch of the checker; off by default."""
    lib().orc_set_sophus_mode(int(bool(on)))


def sophus_mode() -> bool:
    return bool(lib().orc_get_sophus_mode())


def _p(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def sdf_loss_forward(pose_init, sdf_grids, sdf_limits, points, epsilons, padding_scales, clearances, disables):
    """-> potentials [N], potential_grads [N,3], collides [N] (float32)."""
    pose_init, sdf_grids, sdf_limits, points = _f32(pose_init), _f32(sdf_grids), _f32(sdf_limits), _f32(points)
    epsilons, padding_scales, clearances, disables = _f32(epsilons), _f32(padding_scales), _f32(clearances), _f32(disables)
    N, O = points.shape[0], pose_init.shape[0]
    pot = np.zeros(N, np.float32)
    grad = np.zeros((N, 3), np.float32)
    col = np.zeros(N, np.float32)
    rc = lib().orc_sdf_loss_forward(_p(pose_init, C.c_float), _p(sdf_grids, C.c_float), _p(sdf_limits, C.c_float),
                                    _p(points, C.c_float), _p(epsilons, C.c_float), _p(padding_scales, C.c_float),
                                    _p(clearances, C.c_float), _p(disables, C.c_float), C.c_int64(N), C.c_int32(O),
                                    _p(pot, C.c_float), _p(grad, C.c_float), _p(col, C.c_float))
    assert rc == 0
    return pot, grad, col


def fk(robot_blob, joints):
    """joints [B,9] radians -> link poses [B,10,4,4] (center_offset applied), joint origins [B,10,3], axes [B,10,3]."""
    robot_blob, joints = _f64(robot_blob), _f64(joints)
    B = joints.shape[0]
    pose = np.zeros((B, 10, 4, 4))
    org = np.zeros((B, 10, 3))
    ax = np.zeros((B, 10, 3))
    lib().orc_fk_batch(_p(robot_blob, C.c_double), _p(joints, C.c_double), C.c_int64(B), _p(pose, C.c_double),
                       _p(org, C.c_double), _p(ax, C.c_double))
    return pose, org, ax


def config_points(robot_blob, P, q):
    robot_blob, q = _f64(robot_blob), _f64(q)
    x = np.zeros((10, P, 3))
    lib().orc_points_of_config(_p(robot_blob, C.c_double), C.c_int32(P), _p(q, C.c_double), _p(x, C.c_double))
    return x


def smooth_matrices(n, dt, goal_set_proj):
    D = np.zeros((n + 1, n)); A = np.zeros((n, n)); Ainv = np.zeros((n, n))
    lib().orc_smooth_matrices(C.c_int(n), C.c_double(dt), C.c_int(int(goal_set_proj)), _p(D, C.c_double),
                              _p(A, C.c_double), _p(Ainv, C.c_double))
    return D, A, Ainv


def fk_sdf(robot_blob, P, batch, joints, soften_fingers=False):
    """joints [S,C,9] -> potentials [S,C,10,P], grads [S,C,10,P,3], collides [S,C,10,P] (float32)."""
    robot_blob, joints = _f64(robot_blob), _f64(joints)
    S, Cn = joints.shape[:2]
    pot = np.zeros((S, Cn, 10, P), np.float32)
    grad = np.zeros((S, Cn, 10, P, 3), np.float32)
    col = np.zeros((S, Cn, 10, P), np.float32)
    objs = np.ascontiguousarray(batch.objects)
    rc = lib().orc_fk_sdf(_p(robot_blob, C.c_double), C.c_int32(P), objs.ctypes.data_as(C.c_void_p),
                          _p(np.ascontiguousarray(batch.scene_begin, np.int32), C.c_int32), _p(_f32(batch.pool), C.c_float),
                          _p(joints, C.c_double), C.c_int32(S), C.c_int32(Cn), C.c_int32(int(soften_fingers)),
                          _p(pot, C.c_float), _p(grad, C.c_float), _p(col, C.c_float))
    assert rc == 0, rc
    return pot, grad, col


def goalset_cost(robot_blob, P, batch, traj_start, goals, n_remaining, dt, soften_fingers=False, want_potentials=False):
    """traj_start [S,9], goals [S,G,9] -> goal_cost [S,G], collides [S,G], (potentials [S,G,n,10,P])."""
    robot_blob, traj_start, goals = _f64(robot_blob), _f64(traj_start), _f64(goals)
    S, G = goals.shape[:2]
    cost = np.zeros((S, G), np.float32)
    col = np.zeros((S, G), np.float32)
    pots = np.zeros((S, G, n_remaining, 10, P), np.float32) if want_potentials else None
    objs = np.ascontiguousarray(batch.objects)
    pool = _f32(batch.pool)
    rc = lib().orc_goalset_cost(_p(robot_blob, C.c_double), C.c_int32(P), objs.ctypes.data_as(C.c_void_p),
                                _p(np.ascontiguousarray(batch.scene_begin, np.int32), C.c_int32), _p(pool, C.c_float),
                                _p(traj_start, C.c_double), _p(goals, C.c_double), C.c_int32(S), C.c_int32(G),
                                C.c_int32(n_remaining), C.c_double(dt), C.c_int32(int(soften_fingers)),
                                _p(cost, C.c_float), _p(pots, C.c_float), _p(col, C.c_float))
    assert rc == 0, rc
    return (cost, col, pots) if want_potentials else (cost, col)


def chomp_optimize(robot_blob, params: ChompParams, traj, start, end, goal, goal_point, pot, pgrad, col, active=None):
    """S trajectories: traj [S,n,9] (a copy is updated and returned) -> new_traj, grad [S,n,9], cost_traj [S,n], info [S,16]."""
    robot_blob = _f64(robot_blob)
    traj = _f64(traj).copy()
    S, n = traj.shape[:2]
    start, end, goal, goal_point = _f64(start), _f64(end), _f64(goal), _f64(goal_point)
    pot, pgrad, col = _f32(pot), _f32(pgrad), _f32(col)
    grad = np.zeros((S, n, NUM_DOF))
    cost_traj = np.zeros((S, n))
    info = np.zeros((S, INFO_STRIDE))
    act = None if active is None else np.ascontiguousarray(active, np.int32)
    rc = lib().orc_chomp_optimize(_p(robot_blob, C.c_double), C.byref(params), _p(traj, C.c_double), _p(start, C.c_double),
                                  _p(end, C.c_double), _p(goal, C.c_double), _p(goal_point, C.c_double),
                                  _p(pot, C.c_float), _p(pgrad, C.c_float), _p(col, C.c_float), _p(act, C.c_int32),
                                  C.c_int32(S), _p(grad, C.c_double), _p(cost_traj, C.c_double), _p(info, C.c_double))
    assert rc == 0, rc
    return traj, grad, cost_traj, info


def goal_update(params: LearnerParams, traj, goal_set, reach, goal_cost, state):
    """-> goal_idx [S], end [S,9], goal_rows [S,c,9], goal_point [S,9], cost_vector [S,G]; `state` is updated in place."""
    traj, goal_set = _f64(traj), _f64(goal_set)
    S, G, c = traj.shape[0], params.num_goals, params.constraint_num
    reach = None if reach is None else _f64(reach)
    goal_cost = _f32(goal_cost)
    assert state.dtype == np.float64 and state.flags.c_contiguous and state.shape == (S, 7 * G + 10)
    idx = np.zeros(S, np.int32)
    end = np.zeros((S, 9)); rows = np.zeros((S, c, 9)); gp = np.zeros((S, 9)); cv = np.zeros((S, G))
    rc = lib().orc_goal_update(C.byref(params), _p(traj, C.c_double), _p(goal_set, C.c_double), _p(reach, C.c_double),
                               _p(goal_cost, C.c_float), _p(state, C.c_double), C.c_int32(S), _p(idx, C.c_int32),
                               _p(end, C.c_double), _p(rows, C.c_double), _p(gp, C.c_double), _p(cv, C.c_double))
    assert rc == 0, rc
    return idx, end, rows, gp, cv


def point_cloud_sdf(points, origin, resolution, dims):
    """-> float32 [X,Y,Z] nearest-point distance grid (PointEnv.compute_sdf_from_points)."""
    points, origin = _f64(points), _f64(origin)
    dims = np.ascontiguousarray(dims, np.int32)
    out = np.zeros(tuple(int(d) for d in dims), np.float32)
    rc = lib().orc_point_cloud_sdf(_p(points, C.c_double), C.c_int32(points.shape[0]), _p(origin, C.c_double), C.c_double(resolution),
                                   _p(dims, C.c_int32), _p(out, C.c_float))
    assert rc == 0
    return out
