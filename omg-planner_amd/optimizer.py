"""`Optimizer` — host-side mirror of the reference's CHOMP optimiser class (omg/optimizer.py:9-174).

``optimize(traj, force_update=False, info_only=False)`` is ONE device launch pair
(omgx_fk_sdf + omgx_chomp_optimize): schedules -> total loss -> joint-limit check -> projected
A^-1 step -> Trajectory.update -> smooth joint-limit projection, all inside k_chomp_optimize; the
trajectory comes back once.  ``goal_set_projection`` / ``handle_joint_limit`` / ``compute_traj_v`` /
``check_joint_limit`` are kept as small numpy utilities with the reference's signatures for callers that
use them on their own; ``optimize`` does not go through them.
"""
from __future__ import annotations

import time

import numpy as np


class Optimizer(object):
    def __init__(self, scene, cost):
        self.cfg = scene.config
        self.joint_lower_limit = scene.robot.joint_lower_limit
        self.joint_upper_limit = scene.robot.joint_upper_limit
        self.cost = cost
        self.step = 0
        self.time = 0.0
        self.time_elapsed = time.time()

    def report(self, curve, info):
        """Debug text when cfg.report_cost is set (the role of omg/optimizer.py:23-57); returns the lines."""
        if not self.cfg.report_cost:
            return []
        bar = "=" * 49
        lines = [
            bar,
            f"step: {self.step:.2f}, time: {self.time_elapsed:.2f}, lr: {self.cfg.step_size:.5f}, collide: {info['collide']}",
            f"joint limit: {curve.min():.2f}/{np.min(self.joint_lower_limit):.2f}, {curve.max():.2f}/{np.max(self.joint_upper_limit):.2f}, "
            f"violate: {info['violate_limit']} reach: {info['reach']:.2f} timestep {self.cfg.timesteps}",
            f"obs:{info['obs']:.2f}, smooth:{info['smooth']:.2f}, grasp:{info['grasp']:.2f} total:{info['cost']:.2f} ",
            f"obs_grad:{info['weighted_obs_grad']:.2f}, smooth_grad:{info['weighted_smooth_grad']:.2f}, "
            f"grasp_grad:{info['weighted_grasp_grad']:.2f} total_grad:{info['grad']:.2f}",
            bar,
        ]
        print("\n".join(lines))
        return lines

    def update(self):
        """Weight / step-size schedules, written into cfg like the reference (omg/optimizer.py:59-80)."""
        self.step += 1
        self.time_elapsed = time.time() - self.time
        self.time = time.time()
        cfg = self.cfg
        cfg.obstacle_weight = cfg.base_obstacle_weight * cfg.cost_schedule_decay ** self.step
        cfg.smoothness_weight = cfg.smoothness_base_weight * cfg.cost_schedule_boost ** self.step
        cfg.grasp_weight = cfg.base_grasp_weight * cfg.cost_schedule_decay ** self.step
        cfg.step_size = cfg.step_decay_rate ** self.step * cfg.base_step_size

    def reset(self):
        self.step = 0

    def optimize(self, traj, force_update=False, info_only=False):
        """One CHOMP step (omg/optimizer.py:115-135); returns the reference's info dict."""
        self.update()
        cost, cfg = self.cost, self.cfg
        n = traj.data.shape[0]

        def chosen_rows(idx):  # optimizer.py:93-99
            if cfg.use_standoff:
                return np.asarray(cost.target_obj.reach_grasps[idx], np.float64)
            return np.asarray(traj.goal_set[idx], np.float64)[None]
        do_update = 0 if info_only else (1 if force_update else 2)
        loop = getattr(cost, "_loop", None)
        if loop is not None and loop.matches(traj):
            # the planner loop's fast path (device_loop.DeviceLoop): trajectory, goal set and learner state stay on the device; a
            # deferred Learner.update_goal() rides on this call's launches; one download
            if cfg.goal_set_proj:
                rows = lambda: chosen_rows(int(traj.goal_idx))
                point = lambda: np.asarray(traj.goal_set[int(traj.goal_idx)], np.float64)
            else:
                rows = lambda: np.tile(np.asarray(traj.end, np.float64), (cost._params(n, 0).constraint_num, 1))
                point = lambda: np.asarray(traj.end, np.float64)
            st = loop.optimize(traj, cost._params(n, do_update, loop.P), rows, point)  # (loop.matches has just checked the robot's points)
            collision_pts = cost._collision_pts_recompute(traj.data)
        else:
            if cfg.goal_set_proj:
                chosen = chosen_rows(int(traj.goal_idx))
                goal_point = np.asarray(traj.goal_set[int(traj.goal_idx)], np.float64)
            else:
                chosen = np.tile(np.asarray(traj.end, np.float64), (cost._params(n, 0).constraint_num, 1))
                goal_point = np.asarray(traj.end, np.float64)
            # one upload, two launches, one download (cost._Staging); results are host views valid until the next call
            st = cost._run_step(traj.data, traj.start, traj.end, chosen, goal_point, do_update, False)
            collision_pts = cost._collision_pts_builder(traj.data, st)
        i = st.h("info")[0].copy()
        info = {
            "obs": i[1], "smooth": i[2], "grasp": 0, "weighted_obs": i[3], "weighted_smooth": i[4],
            "weighted_smooth_grad": i[6], "weighted_obs_grad": i[5], "weighted_grasp_grad": 0, "weighted_grasp": 0,
            "gradient": st.h("grad")[0].copy(), "failure_terminate": bool(i[11]), "cost": i[0], "grad": i[7],
            "terminate": bool(i[10]), "collide": np.float32(i[8]), "standoff_idx": int(i[13]), "reach": i[9],
            "execute": bool(i[12]), "cost_traj": st.h("cost_traj")[0].copy(), "violate_limit": bool(i[14]),
        }
        from .cost import LazyInfo
        info = LazyInfo(info, collision_pts=collision_pts)  # built on first access (viewer only)
        info["text"] = self.report(np.asarray(traj.data), info)
        if (info["terminate"] and not force_update) or info_only:
            return info
        traj.set(st.h("traj")[0].copy())  # update + handle_joint_limit already applied on the device
        return info

    # ---- numpy utilities with the reference's signatures (not used by optimize) -----------------------
    def goal_set_projection(self, traj, grad):
        """Projected update (omg/optimizer.py:88-113) with the closed-form projector of csrc/omg_chomp.hip."""
        cfg = self.cfg
        if cfg.use_standoff:
            chosen = np.asarray(self.cost.target_obj.reach_grasps[int(traj.goal_idx)])
        else:
            chosen = np.asarray(traj.goal_set[int(traj.goal_idx)])[None]
        c, n = chosen.shape[0], traj.data.shape[0]
        Ag = cfg.Ainv.dot(grad)
        b = traj.data[-c:] - chosen
        M = np.zeros((n, c))
        M[: n - c, 0] = (np.arange(n - c) + 1.0) / (n - c + 1.0)
        M[n - c:, :] = np.eye(c)
        return -cfg.step_size * Ag + cfg.step_size * M.dot(Ag[-c:]) - M.dot(b)

    def compute_traj_v(self, curve):
        low = curve < self.joint_lower_limit
        high = curve > self.joint_upper_limit
        return low * (self.joint_lower_limit - curve) + high * (self.joint_upper_limit - curve)

    def handle_joint_limit(self, curve):
        """Smooth joint-limit projection (omg/optimizer.py:148-164)."""
        cnt = 0
        v = self.compute_traj_v(curve)
        while np.linalg.norm(v) > 1e-2 and cnt < self.cfg.joint_limit_max_steps:
            vs = self.cfg.Ainv.dot(v)
            k = np.unravel_index(np.abs(v).argmax(), v.shape)
            curve = curve + np.abs(v).max() / (np.abs(vs[k]) + 1e-8) * vs
            v = self.compute_traj_v(curve)
            cnt += 1
        return curve

    def check_joint_limit(self, curve, info):
        """Needs BOTH a low and a high violation, like the reference (omg/optimizer.py:166-174)."""
        low = (curve < self.joint_lower_limit - 5e-3).any()
        high = curve > self.joint_upper_limit + 5e-3
        over = bool((low * high).any())
        info["violate_limit"] = over
        info["terminate"] = info["terminate"] and (not over)
