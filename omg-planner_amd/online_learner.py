"""`Learner` — host-side mirror of the reference's online goal-selection class (omg/online_learner.py:61-249) for the
single-scene drop-in level: same constructor, attributes and methods, with `cost_vector` (goal-set obstacle batch, arc-length
weighting, per-goal reduction, distance proxy, normalisation) and the FTL / FTC / Exp / MD / Proj updates running on the
device (omgx_goalset_cost + omgx_goal_update) instead of numpy + one `batch_obstacle_cost` round trip per iteration.

    import sys, omg_planner_amd.online_learner
    sys.modules["omg.online_learner"] = omg_planner_amd.online_learner      # before omg.planner is imported

The batched planner (`engine.ChompEngine`) does not go through this class: it keeps the same state for S scenes on the
device and fuses the update with the optimiser step.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, ops


class Learner(object):
    def __init__(self, env, traj, cost):
        cfg, n = env.config, len(traj.goal_set)
        eta = np.sqrt(np.log(n + 1) / cfg.optim_steps)
        etas = [eta * 2.0 ** k for k in (-2, -1, 0, 2, 4)]
        # The reference's attribute surface (omg/online_learner.py:67-92): planner and user code read these names, so names and
        # initial values are the interface.  The numerical state among them (p, sum_costs, experts_p, q, experts_costs) is the
        # host view of the device state and is refreshed after every update.
        vars(self).update(
            cfg=cfg, env=env, traj=traj, cost=cost, alg_name=cfg.ol_alg, N=n, T=cfg.optim_steps, t=0.0, last_leader=0,
            Ti=np.zeros(n), Tis=[], weights=np.ones(n), p=np.full(n, 1.0) / n, sum_costs=np.zeros(n),
            eta=eta, etas=etas, num_experts=len(etas), delta=np.ones(n) / (4 * n + 1),
            experts_p=[np.ones(n) / n for _ in etas], experts_costs=np.zeros(len(etas)), q=np.ones(len(etas)) / len(etas),
        )
        if self.alg_name not in _lib.ALG:
            raise ValueError(f"cfg.ol_alg = {self.alg_name!r}: the learner knows {sorted(_lib.ALG)}")
        self._dev = cost.device
        self._state = None  # device copy of (sum_costs | p | experts_p | q | experts_costs), created with the goal tensors
        self._goals_key = None
        if self.alg_name != "Proj" and len(self.env.objects[self.env.target_idx].reach_grasps) > 0:  # online_learner.py:96-102
            costs = self.cost_vector()
            self.traj.goal_idx = int(np.argmin(costs))
            self.traj.end = self.traj.goal_set[self.traj.goal_idx]
            self.traj.interpolate_waypoints()

    # ---- device side -------------------------------------------------------------------------------
    def _tensors(self):
        """Goal set / standoff tails / learner state on the device; rebuilt when the goal arrays are replaced."""
        goal_set = np.asarray(self.traj.goal_set, np.float64)
        reach = self.env.objects[self.env.target_idx].reach_grasps
        key = (id(self.traj.goal_set), goal_set.shape, id(reach), bool(self.cfg.use_standoff))
        if self._goals_key != key:
            f64 = dict(dtype=torch.float64, device=self._dev)
            self._goal_set = torch.as_tensor(goal_set[None], **f64).contiguous()
            self._reach = None
            self._cv_goals = self._goal_set
            if self.cfg.use_standoff:
                r = np.asarray(reach, np.float64)
                if r.ndim != 3 or r.shape[0] != goal_set.shape[0]:
                    raise _lib.OmgHipError("cfg.use_standoff needs target_obj.reach_grasps [G,c,9]")
                self._reach = torch.as_tensor(r[None], **f64).contiguous()
                self._cv_goals = self._reach[:, :, -1, :].contiguous()  # online_learner.py:121-125
            G, c = goal_set.shape[0], (self._reach.shape[2] if self._reach is not None else 1)
            if self._state is None or self._state.shape[1] != 7 * G + 10:
                self._state = ops.learner_state(1, G, self._dev)
            self._idx = torch.zeros(1, dtype=torch.int32, device=self._dev)
            self._end = torch.zeros((1, 9), **f64)
            self._rows = torch.zeros((1, c, 9), **f64)
            self._gp = torch.zeros((1, 9), **f64)
            self._cv = torch.zeros((1, G), **f64)
            self._gcost = torch.zeros((1, G), dtype=torch.float32, device=self._dev)
            self._gcol = torch.zeros((1, G), dtype=torch.float32, device=self._dev)
            self._goals_key = key
        return self._goal_set, self._reach, self._cv_goals

    def _params(self, alg: str) -> _lib.LearnerParams:
        cfg = self.cfg
        p = _lib.LearnerParams()
        p.alg = _lib.ALG[alg]
        p.num_goals, p.n_waypoints = self.N, int(cfg.timesteps)
        p.start_idx = min(int((self.t / cfg.optim_steps) * cfg.timesteps), cfg.timesteps - 1)  # online_learner.py:109-110
        p.use_standoff = int(bool(cfg.use_standoff))
        p.constraint_num = int(self._rows.shape[1])
        p.normalize_cost = int(bool(cfg.normalize_cost))
        p.base_obstacle_weight = float(cfg.base_obstacle_weight)
        p.smooth_weight = float(cfg.smoothness_base_weight * cfg.dist_eps)
        p.eta = float(self.eta)
        return p

    def _run(self, alg: str, state: torch.Tensor):
        """cost_vector's device work + one update of `state` with rule `alg`; returns the learner parameters used."""
        goal_set, reach, cv_goals = self._tensors()
        prm = self._params(alg)
        model, robot = self.cost._robot_model()
        traj = torch.as_tensor(np.ascontiguousarray(self.traj.data, np.float64)[None], dtype=torch.float64).to(self._dev)
        if traj.shape[1] != prm.n_waypoints:
            raise _lib.OmgHipError(f"trajectory has {traj.shape[1]} waypoints, cfg.timesteps is {prm.n_waypoints}")
        if alg != "Proj":
            ops.goalset_cost(robot, model.points_per_link, self.cost._scenes(), traj[:, prm.start_idx], cv_goals,
                             prm.n_waypoints - prm.start_idx, float(self.cfg.time_interval), soften_fingers=False,
                             out=(self._gcost, self._gcol))
        ops.goal_update(prm, traj, goal_set, reach, self._gcost, state, self._idx, self._end, self._rows, self._gp, self._cv)
        return prm

    def _pull_state(self):
        G = self.N
        s = self._state[0].cpu().numpy()
        self.sum_costs, self.p = s[:G].copy(), s[G:2 * G].copy()
        self.experts_p = [s[2 * G + i * G: 3 * G + i * G].copy() for i in range(5)]
        self.q, self.experts_costs = s[7 * G:7 * G + 5].copy(), s[7 * G + 5:7 * G + 10].copy()

    # ---- the reference's methods -------------------------------------------------------------------
    def cost_vector(self):
        """Objective cost estimate per goal at the current self.t (online_learner.py:104-160); the state is untouched."""
        cfg = self.cfg
        reach = self.env.objects[self.env.target_idx].reach_grasps
        if getattr(cfg, "traj_init", "grasp") == "grasp" and (len(reach) == 0 or (cfg.use_standoff and len(np.array(reach).shape) == 2)):
            return np.zeros(1)
        self._tensors()
        self._run("FTC", self._state.clone())  # FTC keeps no state: only the cost vector is of interest
        return self._cv[0].cpu().numpy()

    def update_goal_dist(self):
        """One step of the configured rule on the goal distribution (online_learner.py:162-234)."""
        self._tensors()
        self._run(self.alg_name, self._state)
        self._pull_state()
        if self.alg_name in ("FTL", "FTC"):
            self.last_leader = int(self._idx[0])
        if self.alg_name == "Proj":  # online_learner.py:200-210: one-hot on the goal closest to the trajectory's end
            self.p = np.zeros(self.N)
            self.p[int(self._idx[0])] = 1

    def update_goal(self):
        """Take the arg-max of the goal distribution (online_learner.py:237-249); True when the goal changed."""
        self.t += 1
        self.update_goal_dist()
        goal_idx_old = self.traj.goal_idx
        self.traj.goal_idx = int(self._idx[0])  # np.argmax(self.p) with numpy's NaN / tie rules, taken on the device
        self.traj.end = self.traj.goal_set[self.traj.goal_idx]
        self.Ti[self.traj.goal_idx] += 1
        self.Tis.append(self.Ti)
        return self.traj.goal_idx != goal_idx_old
