"""`Cost` — host-side mirror of the reference's cost class surface (omg/cost.py:12-532), backed by the
HIP kernels of libomg_hip.so.

What callers of the reference touch (grep over omg/planner.py, online_learner.py, core.py):
``Cost(env)``, the attributes ``env / cfg / target_obj / sdf_loss``, ``batch_obstacle_cost(...)`` and
``compute_total_loss(traj)`` (through ``Optimizer.optimize``).  Those run on the device:

===============================  ==========================================================
method                           device entry point (include/omg_hip.h)
===============================  ==========================================================
compute_obstacle_cost_layer      omgx_sdf_loss_forward (the SDF layer on explicit points)
batch_obstacle_cost              omgx_fk_sdf (FK -> points -> SDF, optional arc-length weights)
compute_total_loss               omgx_fk_sdf + omgx_chomp_optimize (info only)
compute_collision_loss           same launch, un-weighted pieces from the `aux` buffer
compute_smooth_loss              same launch, un-weighted pieces from the `aux` buffer
forward_poses                    omgx_forward_kinematics
===============================  ==========================================================

``env`` is duck-typed like the reference's ``Env`` (omg/core.py:239-411): ``env.robot`` with
``collision_points [10,P,3]``, ``joint_lower_limit / joint_upper_limit [1,9]`` and (optionally)
``robot_kinematics`` carrying the ``_pose_0 / _tip2joint / _joint_axis / center_offset`` tables;
``env.objects[i]`` with ``name / pose_mat / attached``; ``env.target_idx``; ``env.sdf_torch [O,X,Y,Z]`` and
``env.sdf_limits [O,10]`` device tensors (used in place, never copied); ``env.config``.

The remaining small methods (``forward_points``, ``color_point``) are array plumbing for visualisation,
kept in numpy exactly because they are not on the hot path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops
from . import scenes as sc
from .robot import PandaModel


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class SDFLoss:
    """Callable with the signature of layers/sdf_matching_loss.py:SDFLoss.forward."""

    def __call__(self, pose_init, sdf_grids, sdf_limits, points, epsilons, padding_scales, clearances, disables):
        return tuple(ops.sdf_loss_forward(pose_init, sdf_grids, sdf_limits, points, epsilons, padding_scales,
                                          clearances, disables))

    forward = __call__


class LazyInfo(dict):
    """The info dict of compute_total_loss / Optimizer.optimize.  "collision_pts" (the [n,10,p,12] visualisation array read
    by omg/core.py:561-570,661) is built on first access: it needs one more launch, a copy and numpy work that an
    optimisation loop without a viewer never uses.  info["collision_pts"] works as in the reference; before that access the
    key is absent from keys() / len()."""

    def __init__(self, *a, collision_pts=None, **k):
        super().__init__(*a, **k)
        self._collision_pts = collision_pts

    def __missing__(self, key):
        if key == "collision_pts" and self._collision_pts is not None:
            self[key] = self._collision_pts()
            self._collision_pts = None
            return self[key]
        raise KeyError(key)


class LazyArray:
    """The `vis_pts` array of batch_obstacle_cost, built on first use.  Learner.cost_vector takes only element [0] of the
    returned tuple (online_learner.py:134-141) and must not pay for 27 MB of visualisation data per call; a viewer that does
    read it (indexing, `.shape`, `np.asarray`, arithmetic through numpy) gets the reference's array."""

    def __init__(self, shape, build):
        self.shape, self.dtype, self.ndim = tuple(shape), np.dtype(np.float64), len(shape)
        self._build, self._value = build, None

    def _get(self):
        if self._value is None:
            self._value = self._build()
            self._build = None
            assert self._value.shape == self.shape
        return self._value

    def __array__(self, dtype=None, copy=None):
        v = self._get()
        return v if dtype is None else v.astype(dtype)

    def __getitem__(self, idx):
        return self._get()[idx]

    def __setitem__(self, idx, value):
        self._get()[idx] = value

    def __len__(self):
        return self.shape[0]

    def __getattr__(self, name):  # anything else an ndarray offers (reshape, sum, copy, ...)
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self._get(), name)


class _Staging:
    """Buffers of ONE single-trajectory step (Optimizer.optimize / compute_total_loss): a device buffer holding every input
    and output of omgx_fk_sdf + omgx_chomp_optimize back to back and a pinned host mirror, so that a call costs one
    host->device copy (start | end | goal rows | goal point | trajectory), two launches and one device->host copy
    (trajectory | grad | cost_traj | info | aux | potentials | gradients | collisions) instead of five + six copies."""

    def __init__(self, n: int, c: int, P: int, aux_doubles: int, device):
        f64 = [("start", (1, 9)), ("end", (1, 9)), ("goal", (1, c, 9)), ("goal_point", (1, 9)), ("traj", (1, n, 9)),
               ("grad", (1, n, 9)), ("cost_traj", (1, n)), ("info", (1, _lib.INFO_STRIDE)), ("aux", (1, aux_doubles))]
        f32 = [("pot", (1, n, 10, P)), ("pgrad", (1, n, 10, P, 3)), ("col", (1, n, 10, P))]
        off, self.slots = 0, {}
        for name, shape in f64:
            self.slots[name] = (off, shape, torch.float64, np.float64)
            off += int(np.prod(shape)) * 8
        for name, shape in f32:
            self.slots[name] = (off, shape, torch.float32, np.float32)
            off += int(np.prod(shape)) * 4
        self.nbytes = (off + 7) & ~7
        self.in_end = self.slots["grad"][0]     # inputs: everything up to and including the trajectory
        self.out_begin = self.slots["traj"][0]  # outputs: from the trajectory (updated in place) to the end
        self.dev = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.host = torch.zeros(self.nbytes, dtype=torch.uint8).pin_memory()
        hnp = self.host.numpy()
        self._d = {k: self.dev[off: off + int(np.prod(shape)) * tdt.itemsize].view(tdt).view(shape)
                   for k, (off, shape, tdt, _) in self.slots.items()}
        self._h = {k: hnp[off: off + int(np.prod(shape)) * np.dtype(ndt).itemsize].view(ndt).reshape(shape)
                   for k, (off, shape, _, ndt) in self.slots.items()}
        self._d_in, self._h_in = self.dev[: self.in_end], self.host[: self.in_end]
        self._d_out, self._h_out = self.dev[self.out_begin:], self.host[self.out_begin:]

    def d(self, name):  # device view
        return self._d[name]

    def h(self, name):  # host (numpy) view of the pinned mirror: valid until the next call
        return self._h[name]

    def upload(self):
        self._d_in.copy_(self._h_in, non_blocking=True)

    def download(self):
        self._h_out.copy_(self._d_out, non_blocking=True)
        torch.cuda.current_stream(self.dev.device).synchronize()


class Cost(object):
    """Obstacle and smoothness cost + gradients of a trajectory (omg/cost.py:12-16)."""

    def __init__(self, env):
        self.env = env
        self.cfg = env.config
        self.sdf_loss = SDFLoss()
        if len(self.env.objects) > 0:
            self.target_obj = self.env.objects[self.env.target_idx]
        self.device = env.sdf_torch.device if isinstance(getattr(env, "sdf_torch", None), torch.Tensor) else torch.device("cuda:0")
        self._model = None
        self._robot = None
        self._points_version = None
        self._staging = {}

    # -- device-side state -------------------------------------------------------------------------
    def _robot_model(self):
        """Robot blob (kinematic tables + current collision points); rebuilt when the points change
        (Robot.resample_attached_object_collision_points swaps them, omg/core.py:192-236)."""
        pts = np.asarray(self.env.robot.collision_points, dtype=np.float64)
        key = (pts.shape, pts.tobytes())
        if self._points_version != key:
            m = PandaModel(collision_points=pts)
            kin = getattr(self.env.robot, "robot_kinematics", None)
            if kin is not None and hasattr(kin, "_pose_0"):
                m.pose_0 = np.ascontiguousarray(kin._pose_0, np.float64)
                m.tip2joint = np.ascontiguousarray(kin._tip2joint, np.float64)
                m.joint_axis = np.ascontiguousarray(kin._joint_axis, np.float64)
                m.center_offset = np.ascontiguousarray(kin.center_offset, np.float64)
            m.joint_lower_limit = np.asarray(self.env.robot.joint_lower_limit, np.float64).reshape(1, 9).copy()
            m.joint_upper_limit = np.asarray(self.env.robot.joint_upper_limit, np.float64).reshape(1, 9).copy()
            self._model, self._robot, self._points_version = m, ops.robot_blob(m, self.device), key
        return self._model, self._robot

    def _layer_params(self):
        """Per-object parameters of compute_obstacle_cost_layer (cost.py:303-328), numpy float32."""
        scene = sc.Scene([sc.SceneObject(o.name, np.asarray(o.pose_mat), None, bool(getattr(o, "attached", False)))
                          for o in self.env.objects], int(self.env.target_idx))
        return sc.layer_params(scene, epsilon=self.cfg.epsilon, target_epsilon=self.cfg.target_epsilon,
                               clearance=self.cfg.clearance, target_clearance=self.cfg.target_clearance,
                               disable_collision_set=tuple(self.cfg.disable_collision_set))

    def invalidate(self) -> None:
        """Forget the cached object table and influence boxes.  They are keyed on the identity and the in-place version
        counter of env.sdf_torch / env.sdf_limits, which a write through a raw device pointer (the C ABI, a custom kernel) does
        not advance: call this after such a write — or bump the counter, torch.autograd.graph.increment_version(env.sdf_torch) —
        otherwise the culling boxes of the old volume are applied to the new one."""
        self._scenes_cache = None
        self._pool_ref = None

    def _scenes(self) -> ops.DeviceScenes:
        """One-scene object table addressing env.sdf_torch IN PLACE (rebuilt per call like the reference
        rebuilds its five parameter tensors per call; object poses may have changed)."""
        # Nothing to rebuild while the inputs are the same as in the previous call (the usual case inside an optimisation
        # loop): object names / poses / attached flags, target, layer parameters of cfg, and the SDF tensors (same tensor
        # objects at the same in-place version).
        cfg, env = self.cfg, self.env
        lim_t = env.sdf_limits
        key = (id(env.sdf_torch), env.sdf_torch._version, id(lim_t), getattr(lim_t, "_version", None), int(env.target_idx),
               float(cfg.epsilon), float(cfg.target_epsilon), float(cfg.clearance), float(cfg.target_clearance),
               tuple(cfg.disable_collision_set),
               tuple((o.name, bool(getattr(o, "attached", False)), np.asarray(o.pose_mat, np.float64).tobytes()) for o in env.objects))
        cached = getattr(self, "_scenes_cache", None)
        if cached is not None and cached[0] == key and cached[1]() is env.sdf_torch and cached[3]() is lim_t:  # same OBJECTS, not recycled ids
            return cached[2]
        poses, eps, pad, clr, dis = self._layer_params()
        limits = _np(self.env.sdf_limits).astype(np.float32)
        table = sc.table_from_padded(poses, limits, eps, pad, clr, dis)
        # shrink the far boxes to the voxels that can contribute (same results, fewer exact lookups); the ranges are
        # cached per (volume, epsilon, clearance) on a host copy of the volumes taken once per env.sdf_torch
        # (same tensor OBJECT and same in-place version counter: a rebuilt env.sdf_torch may reuse the old address)
        t = self.env.sdf_torch
        ref = getattr(self, "_pool_ref", None)
        if ref is None or ref() is not t or self._pool_version != t._version:
            import weakref
            self._pool_ref, self._pool_version = weakref.ref(t), t._version
            self._pool_host, self._infl_cache = t.detach().reshape(-1).cpu().numpy(), {}
        sc.tighten_far_boxes(table, self._pool_host, self._infl_cache)
        ds = ops.DeviceScenes.__new__(ops.DeviceScenes)
        ds.device = self.device
        ds.num_scenes = 1
        ds.objects = torch.from_numpy(table.view(np.uint8).copy()).to(self.device)
        ds.scene_begin = torch.tensor([0, len(table)], dtype=torch.int32, device=self.device)
        ds.pool = self.env.sdf_torch.reshape(-1)
        if not (ds.pool.is_cuda and ds.pool.dtype == torch.float32 and ds.pool.is_contiguous()):
            raise _lib.OmgHipError("env.sdf_torch must be a contiguous float32 device tensor")
        import weakref
        self._scenes_cache = (key, weakref.ref(env.sdf_torch), ds, weakref.ref(lim_t))
        return ds

    def _params(self, n: int, do_update: int, P: "int | None" = None) -> _lib.ChompParams:
        """omgx_chomp_params of the moment.  The fields that change from call to call (the optimiser's schedule, do_update) are
        written every time; everything else is kept while cfg says the same (a tuple comparison instead of ~40 attribute
        writes through ctypes: 10 us of a 90 us planner iteration)."""
        cfg = self.cfg
        lsw = cfg.link_smooth_weight
        key = (n, self._robot_model()[0].points_per_link if P is None else P, cfg.top_k_collision, cfg.consider_finger, cfg.goal_set_proj, cfg.use_standoff,
               cfg.reach_tail_length, cfg.uncheck_finger_collision, cfg.joint_limit_max_steps, cfg.allow_collision_point, cfg.pre_terminate,
               cfg.time_interval, cfg.clip_grad_scale, cfg.terminate_smooth_loss, lsw if isinstance(lsw, (int, float)) else tuple(np.ravel(lsw)))
        cached = self.__dict__.get("_params_cache")
        if cached is None or cached[0] != key:
            p = _lib.ChompParams()
            p.n_waypoints, p.n_points = n, key[1]
            p.top_k = int(cfg.top_k_collision)
            p.consider_finger = int(cfg.consider_finger)
            p.goal_set_proj = int(cfg.goal_set_proj)
            p.use_standoff = int(cfg.use_standoff)
            p.constraint_num = int(cfg.reach_tail_length) if cfg.use_standoff else 1
            p.uncheck_finger_collision = int(cfg.uncheck_finger_collision)
            p.joint_limit_max_steps = int(cfg.joint_limit_max_steps)
            p.allow_collision_point = int(cfg.allow_collision_point)
            p.pre_terminate = int(cfg.pre_terminate)
            p.time_interval = float(cfg.time_interval)
            p.clip_grad_scale = float(cfg.clip_grad_scale)
            p.terminate_smooth_loss = float(cfg.terminate_smooth_loss)
            w = np.broadcast_to(np.asarray(cfg.link_smooth_weight, np.float64).ravel(), (9,))
            for d in range(9):
                p.link_smooth_weight[d] = float(w[d])
            self._params_cache = cached = (key, p)
        p = _lib.ChompParams.from_buffer_copy(cached[1])  # a fresh struct: callers keep theirs across later calls
        p.do_update = do_update
        p.obstacle_weight = float(cfg.obstacle_weight)
        p.smoothness_weight = float(cfg.smoothness_weight)
        p.step_size = float(cfg.step_size)
        return p

    def _t(self, a, dtype=torch.float64):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    # -- kinematics helpers --------------------------------------------------------------------------
    def forward_poses(self, joints):
        """Link poses [10,4,4], joint origins [10,3] and axes [10,3] of ONE configuration given in DEGREES
        with the dummy hand joint (the output of wrap_value), as omg/cost.py:45-58."""
        model, robot = self._robot_model()
        j = np.asarray(joints, np.float64)
        q = (np.concatenate([j[:7], j[8:10]]) if j.shape[0] > 9 else j) / 180.0 * np.pi  # deg2rad as robot_pykdl.py:70-73 writes it
        poses, org, ax = ops.forward_kinematics(robot, model.points_per_link, self._t(q[None]))
        return _np(poses[0]), _np(org[0]), _np(ax[0])

    def forward_points(self, pose, pts, normals=None):
        """x = R pts + t, returned as [p, links, n, 3] (omg/cost.py:60-72)."""
        x = np.matmul(pose[..., :3, :3], pts[None, ...]) + pose[..., :3, [3]]
        if normals is not None:
            x = np.concatenate([x, np.matmul(pose[..., :3, :3], normals[None, ...])], 2)
        return x.transpose([3, 1, 0, 2])

    def _vis_points(self, q):
        """World positions of the collision points of configurations q [B,9] as the SDF layer receives them: [B,10,p,3]
        float32 (omg/cost.py:209-218: numpy FK + forward_points, then .float())."""
        model, robot = self._robot_model()
        poses = _np(ops.forward_kinematics(robot, model.points_per_link, self._t(np.asarray(q, np.float64).reshape(-1, 9)),
                                           want_joint_info=False)[0])
        pts = np.asarray(self.env.robot.collision_points, np.float64).transpose([0, 2, 1])
        return self.forward_points(poses, pts).transpose([2, 1, 0, 3]).astype(np.float32)

    def _vis_array(self, q, pot, pgrad, col, highlight_top_k: bool):
        """vis_pts [B,10,p,12] as compute_obstacle_cost_layer + color_point fill it (cost.py:355-358, 74-90): columns 0:3
        positions, 6:9 colours from the relative potential (red where colliding), 9:12 gradients; compute_collision_loss
        then paints the top-k points (cost.py:390-399, same argsort expression on the same float32 potentials)."""
        vis = np.zeros([pot.shape[0], 10, pot.shape[2], 12])
        vis[..., :3] = self._vis_points(q)
        vis[..., 6] = pot
        vis[..., 9:] = pgrad
        self.color_point(vis, col)
        k = int(self.cfg.top_k_collision)
        if highlight_top_k and k > 0:
            topk = np.unravel_index(np.argsort(pot.flatten()), pot.shape)
            vis[topk[0][-k:], topk[1][-k:], topk[2][-k:], 6:9] = [235, 52, 195]
        return vis

    def color_point(self, vis_pts, collide):
        """Visualisation colours from relative potential (omg/cost.py:74-90)."""
        pmax = np.amax(vis_pts[..., 6], axis=(-2, -1))[..., None, None]
        pmin = np.amin(vis_pts[..., 6], axis=(-2, -1))[..., None, None]
        vis_pts[..., 6] = 255 * ((vis_pts[..., 6] - pmin) / ((pmax - pmin + 1e-8) + 1e-8))
        vis_pts[..., 7] = 255 - vis_pts[..., 6]
        vis_pts[_np(collide).astype(bool), 6:9] = 255, 0, 0

    # -- reference helpers kept for callers that want the intermediate tensors (not used by compute_total_loss,
    #    which computes the same quantities inside k_chomp_optimize) ------------------------------------------
    def functional_grad(self, v, a, JT, ws_cost, ws_grad):
        """CHOMP workspace functional gradient of points (omg/cost.py:24-43): cost = sum c ||v||,
        grad = J . (||v|| P grad_c - c P a / (||v||^2 + 1e-8)) with P = I - vhat vhat^T."""
        speed = np.linalg.norm(v, axis=-1, keepdims=True)
        vhat = v / (speed + 1e-8)
        P = np.eye(3) - vhat[..., :, None] * vhat[..., None, :]
        kappa = ws_cost[..., None] * np.einsum("...ij,...j->...i", P, a) / (speed ** 2 + 1e-8)
        g = speed * np.einsum("...ij,...j->...i", P, ws_grad) - kappa
        return np.sum(ws_cost * speed[..., 0], axis=-1), np.einsum("...kj,...j->...k", JT, g)

    def compute_point_jacobian(self, joint_origin, x, joint_axis, potentials, type="revolute"):
        """Per-point Jacobians [n, p, joints, 6] of one link (omg/cost.py:92-110); x is [p, n, 3]."""
        xt = np.transpose(x, (1, 0, 2))[:, :, None, :]
        J = np.zeros([xt.shape[0], xt.shape[1], joint_axis.shape[1], 6])
        J[..., :3] = np.cross(joint_axis[:, None], xt - joint_origin[:, None])
        J[..., 3:] = joint_axis[:, None]
        if type == "prsimatic":  # (sic) finger joint: pure translation along its axis
            J[..., -1, :3] = joint_axis[:, [-1], :]
            J[..., -1, 3:] = 0
        return J

    def forward_kinematics_obstacle(self, xi, start, end, arc_length=True):
        """x, v, a [n,10,p,3], Js, potentials, potential_grads, vis_pts, collide of a trajectory (omg/cost.py:112-190):
        FK and the SDF layer on the device, the finite differences / Jacobians with the helpers above."""
        from .util import wrap_joint
        model, robot = self._robot_model()
        P = model.points_per_link
        xi = np.asarray(xi, np.float64)
        n = xi.shape[0]
        q = self._t(np.concatenate([xi, np.asarray(start)[None], np.asarray(end)[None]], 0))
        poses, org, ax = (_np(t) for t in ops.forward_kinematics(robot, P, q))
        pts = np.asarray(self.env.robot.collision_points, np.float64).transpose([0, 2, 1])
        ws = self.forward_points(poses[:n], pts)  # [p, 10, n, 3]
        pot, grad, col = ops.fk_sdf(robot, P, self._scenes(), self._t(xi[None]), soften_fingers=self.cfg.uncheck_finger_collision == -1)
        potentials, potential_grads, collide = _np(pot[0]), _np(grad[0]), _np(col[0])
        vis_pts = np.zeros([n, 10, P, 12])  # n x (m + 1) x p x 12 with m = xi.shape[1] = 9 (cost.py:119-121)
        vis_pts[..., :3] = ws.transpose([2, 1, 0, 3]).astype(np.float32)
        vis_pts[..., 6] = potentials
        vis_pts[..., 9:] = potential_grads
        self.color_point(vis_pts, collide)
        Js = [self.compute_point_jacobian(org[:n][:, wrap_joint(j + 1)], ws[:, j], ax[:n][:, wrap_joint(j + 1)], potentials[:, j],
                                          "prsimatic" if j >= 8 else "revolute") for j in range(10)]
        if not arc_length:
            return Js, potentials, potential_grads, collide.sum()
        ws_start = self.forward_points(poses[n][None], pts)[:, :, 0]
        ws_end = self.forward_points(poses[n + 1][None], pts)[:, :, 0]
        v = self.cfg.get_derivative(ws, ws_start, ws_end, 1).transpose([2, 1, 0, 3])
        a = self.cfg.get_derivative(ws, ws_start, ws_end, 2).transpose([2, 1, 0, 3])
        return ws.transpose([2, 1, 0, 3]), v, a, Js, potentials, potential_grads, vis_pts, collide.sum()

    # -- SDF layer -----------------------------------------------------------------------------------
    def compute_obstacle_cost_layer(self, ws_positions, vis_pts=None, special_check_id=0, uncheck_finger_collision=-1,
                                    grad_free=True):
        """SDF layer on explicit workspace points [n, m, p, 3] (device tensor) — omg/cost.py:288-360."""
        n, m, p, _ = ws_positions.shape
        points = ws_positions.reshape([-1, 3]).contiguous().float()
        poses, eps, pad, clr, dis = self._layer_params()
        dev = points.device
        potentials, potential_grads, collides = self.sdf_loss(
            torch.from_numpy(poses).to(dev), self.env.sdf_torch, self.env.sdf_limits, points, torch.from_numpy(eps).to(dev),
            torch.from_numpy(pad).to(dev), torch.from_numpy(clr).to(dev), torch.from_numpy(dis).to(dev))
        potentials = potentials.reshape([n, m, p])
        potential_grads = potential_grads.reshape([n, m, p, 3])
        collides = collides.reshape([n, m, p])
        if uncheck_finger_collision == -1:  # cost.py:350-353
            potentials[:, -2:] *= 0.1
            potential_grads[:, -2:] *= 0.1
            collides[:, -2:] = 0
        if vis_pts is not None:
            vis_pts[:, :m, :, :3] = _np(points.reshape([n, m, p, 3]))
            vis_pts[:, :m, :, 6] = _np(potentials)
            vis_pts[:, :m, :, 9:] = _np(potential_grads)
        return potentials, potential_grads, collides

    def batch_obstacle_cost(self, joints, arc_length=-1, only_collide=False, special_check_id=0,
                            uncheck_finger_collision=-1, start=None, end=None, want_vis=True):
        """Obstacle cost of a batch of configurations joints [B,9] — omg/cost.py:192-286.  FK, the point
        transform, the SDF layer and the arc-length weighting all run on the device (omgx_fk_sdf).
        Returns (potentials [B,m,p], grad [B,m,p,3], vis_pts, collide) like the reference; ``want_vis=False``
        skips the [B,m,p,12] host array (5.5 GB at 100 scenes x 128 goals)."""
        model, robot = self._robot_model()
        P = model.points_per_link
        q = self._t(np.asarray(joints, np.float64).reshape(1, -1, 9))
        B = q.shape[1]
        arc = int(arc_length) if arc_length is not None and arc_length > 0 else 0
        pot, grad, col = ops.fk_sdf(robot, P, self._scenes(), q, soften_fingers=uncheck_finger_collision == -1,
                                    arc_length=arc, arc_start=self._t(np.asarray(start, np.float64).reshape(1, 9)) if arc else None,
                                    dt=float(self.cfg.time_interval))
        potentials, grad, collide = pot[0], grad[0], col[0]
        vis_pts = None
        if want_vis:  # coloured from the potentials BEFORE the arc-length weighting (cost.py:219-230); built when first read
            soft, scenes_now = uncheck_finger_collision == -1, self._scenes()

            def build(q=q, grad=grad, collide=collide, plain=None if arc else potentials):
                if plain is None:
                    plain = ops.fk_sdf(robot, P, scenes_now, q, soften_fingers=soft, want_grad=False, want_col=False)[0][0]
                return self._vis_array(_np(q[0]), _np(plain), _np(grad), _np(collide), False)
            vis_pts = LazyArray((B, 10, P, 12), build)
        if only_collide:  # cost.py:279-284
            thr = 0.5 * (self.cfg.epsilon - self.cfg.clearance) ** 2 / self.cfg.epsilon
            potentials = potentials * (potentials > thr).any()
        return potentials, grad, vis_pts, collide

    # -- trajectory losses -----------------------------------------------------------------------------
    def _stage(self, n: int, c: int) -> _Staging:
        P = self._robot_model()[0].points_per_link
        key = (n, c, P)
        st = self._staging.get(key)
        if st is None:
            st = self._staging[key] = _Staging(n, c, P, int(_lib.lib().omgx_chomp_aux_doubles(n)), self.device)
        return st

    def _run_step(self, xi, start, end, goal_rows, goal_point, do_update: int, want_aux: bool) -> _Staging:
        """omgx_fk_sdf + omgx_chomp_optimize for ONE trajectory through the staging buffers; the results are in the
        returned object's host views (st.h(name)) until the next call."""
        model, robot = self._robot_model()
        P = model.points_per_link
        xi = np.asarray(xi, np.float64)
        n = xi.shape[0]
        prm = self._params(n, do_update)
        goal_rows = np.asarray(goal_rows, np.float64).reshape(-1, 9)
        if goal_rows.shape[0] != prm.constraint_num:
            raise _lib.OmgHipError(f"chosen goal has {goal_rows.shape[0]} rows, cfg implies {prm.constraint_num}")
        st = self._stage(n, prm.constraint_num)
        st.h("traj")[0] = xi
        st.h("start")[0] = np.asarray(start, np.float64)
        st.h("end")[0] = np.asarray(end, np.float64)
        st.h("goal")[0] = goal_rows
        st.h("goal_point")[0] = np.asarray(goal_point, np.float64)
        with torch.cuda.device(self.device):
            st.upload()
            ops.fk_sdf(robot, P, self._scenes(), st.d("traj"), soften_fingers=self.cfg.uncheck_finger_collision == -1,
                       out=(st.d("pot"), st.d("pgrad"), st.d("col")))
            ops.chomp_optimize(robot, prm, st.d("traj"), st.d("start"), st.d("end"), st.d("goal"), st.d("goal_point"),
                               st.d("pot"), st.d("pgrad"), st.d("col"), out=(st.d("grad"), st.d("cost_traj"), st.d("info")),
                               aux=st.d("aux") if want_aux else None)
            st.download()
        return st

    def _collision_pts_builder(self, xi, st: _Staging):
        """Closure that builds info["collision_pts"] from copies of this call's layer outputs (the staging views are reused)."""
        xi, pot, pgrad, col = np.array(xi, np.float64), st.h("pot")[0].copy(), st.h("pgrad")[0].copy(), st.h("col")[0].copy()
        return lambda: self._vis_array(xi, pot, pgrad, col, True)

    def _collision_pts_recompute(self, xi):
        """info["collision_pts"] for a call whose layer outputs stayed on the device (device_loop.DeviceLoop): the layer of THAT
        trajectory is evaluated again when somebody asks for the visualisation array."""
        xi = np.array(xi, np.float64)

        def build():
            model, robot = self._robot_model()
            pot, pgrad, col = ops.fk_sdf(robot, model.points_per_link, self._scenes(), self._t(xi[None]),
                                         soften_fingers=self.cfg.uncheck_finger_collision == -1)
            return self._vis_array(xi, _np(pot)[0], _np(pgrad)[0], _np(col)[0], True)
        return build

    def _evaluate(self, xi, start, end, goal_point=None, want_aux=False) -> _Staging:
        """One info-only k_chomp_optimize launch for a single trajectory."""
        n = np.asarray(xi).shape[0]
        c = self._params(n, 0).constraint_num
        end = np.asarray(end, np.float64)
        return self._run_step(xi, start, end, np.tile(end, (c, 1)), end if goal_point is None else goal_point, 0, want_aux)

    def compute_collision_loss(self, xi, start, end):
        """-> obs_cost [n, 10], obs_grad [n, 9], vis_pts, collide  (omg/cost.py:362-423)."""
        st = self._evaluate(xi, start, end, want_aux=True)
        a, pot = st.h("aux")[0].copy(), st.h("pot")[0].copy()
        n = pot.shape[0]
        collide_sum = np.float32(st.h("info")[0, 8])
        build = self._collision_pts_builder(xi, st)  # copies of this call's layer outputs; the array is built when first read
        vis_pts = LazyArray((n, 10, pot.shape[-1], 12), build)
        return a[n * 9: n * 19].reshape(n, 10), a[: n * 9].reshape(n, 9), vis_pts, collide_sum

    def compute_smooth_loss(self, xi, start, end):
        """-> smoothness_loss [n+1], smoothness_grad [n, 9]  (omg/cost.py:425-449)."""
        st = self._evaluate(xi, start, end, want_aux=True)
        a, n = st.h("aux")[0], np.asarray(xi).shape[0]
        return a[n * 28: n * 28 + n + 1].copy(), a[n * 19: n * 28].reshape(n, 9).copy()

    def compute_total_loss(self, traj):
        """-> cost, grad [n, 9], info (the 20 keys of omg/cost.py:509-530)."""
        gp = traj.goal_set[traj.goal_idx] if self.cfg.goal_set_proj and len(traj.goal_set) > 0 else traj.end
        st = self._evaluate(traj.data, traj.start, traj.end, goal_point=gp)
        i = st.h("info")[0].copy()
        grad = st.h("grad")[0].copy()
        pot = st.h("pot")[0]
        n = pot.shape[0]
        cfg = self.cfg
        # compute_total_loss's own flag does not know about joint limits (check_joint_limit amends it later,
        # optimizer.py:166-174); rebuild it from the kernel's numbers (cost.py:489-494)
        terminate = bool((i[8] <= cfg.allow_collision_point) and cfg.pre_terminate and (i[9] < 0.01)
                         and (i[2] < cfg.terminate_smooth_loss))
        info = {
            "obs": i[1], "smooth": i[2], "grasp": 0, "weighted_obs": i[3], "weighted_smooth": i[4],
            "weighted_smooth_grad": i[6], "weighted_obs_grad": i[5], "weighted_grasp_grad": 0, "weighted_grasp": 0,
            "gradient": grad, "failure_terminate": bool(i[11]), "cost": i[0], "grad": i[7], "terminate": terminate,
            "collide": np.float32(i[8]), "standoff_idx": int(i[13]), "reach": i[9], "execute": bool(i[12]),
            "cost_traj": st.h("cost_traj")[0].copy(),
        }
        info = LazyInfo(info, collision_pts=self._collision_pts_builder(traj.data, st))
        return info["cost"], grad, info
