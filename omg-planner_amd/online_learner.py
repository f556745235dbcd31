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

from . import _lib, ops


def _host_value(name):
    """A learner attribute whose truth lives on the device between calls (p, sum_costs, experts_p, q, experts_costs): reading it
    applies a deferred update_goal() and pulls the state if the device is ahead."""
    def get(self):
        loop = self.__dict__.get("_loop")
        if loop is not None:
            if loop._pending is not None:
                loop.flush()
            if loop.state_dirty:
                self._pull_state()
        return self.__dict__["_hv"][name]

    def put(self, value):
        self.__dict__.setdefault("_hv", {})[name] = value
    return property(get, put)


class Learner(object):
    p, sum_costs, experts_p, q, experts_costs = (_host_value(k) for k in ("p", "sum_costs", "experts_p", "q", "experts_costs"))
    # Learner.update_goal() defers its device work to the Optimizer.optimize() that follows (device_loop.DeviceLoop): one fused pair
    # of launches and one download per planner iteration.  False: every call does its own work at once (round-2 behaviour).
    DEFER_UPDATE = True

    def __init__(self, env, traj, cost):
        cfg, n = env.config, len(traj.goal_set)
        eta = np.sqrt(np.log(n + 1) / cfg.optim_steps)
        etas = [eta * 2.0 ** k for k in (-2, -1, 0, 2, 4)]
        # The reference's attribute surface (omg/online_learner.py:67-92): planner and user code read these names, so names and
        # initial values are the interface.  The numerical state among them (p, sum_costs, experts_p, q, experts_costs) is the
        # host view of the device state and is refreshed when it is read after an update.
        self._loop = None
        vars(self).update(
            cfg=cfg, env=env, traj=traj, cost=cost, alg_name=cfg.ol_alg, N=n, T=cfg.optim_steps, t=0.0, last_leader=0,
            Ti=np.zeros(n), Tis=[], weights=np.ones(n),
            eta=eta, etas=etas, num_experts=len(etas), delta=np.ones(n) / (4 * n + 1),
        )
        self.p, self.sum_costs = np.full(n, 1.0) / n, np.zeros(n)
        self.experts_p, self.experts_costs, self.q = [np.ones(n) / n for _ in etas], np.zeros(len(etas)), np.ones(len(etas)) / len(etas)
        if self.alg_name not in _lib.ALG:
            raise ValueError(f"cfg.ol_alg = {self.alg_name!r}: the learner knows {sorted(_lib.ALG)}")
        self._dev = cost.device
        self._goals_key = None
        if self.alg_name != "Proj" and len(self.env.objects[self.env.target_idx].reach_grasps) > 0:  # online_learner.py:96-102
            costs = self.cost_vector()
            self.traj.goal_idx = int(np.argmin(costs))
            self.traj.end = self.traj.goal_set[self.traj.goal_idx]
            self.traj.interpolate_waypoints()

    # ---- device side -------------------------------------------------------------------------------
    def _tensors(self):
        """The device loop of this (learner, trajectory) pair — goal set / standoff tails / learner state / trajectory on the
        device; rebuilt when the goal arrays are replaced (the learner's state is kept while the goal count stays)."""
        from .device_loop import DeviceLoop
        reach = self.env.objects[self.env.target_idx].reach_grasps
        key = (id(self.traj.goal_set), np.asarray(self.traj.goal_set).shape, id(reach), bool(self.cfg.use_standoff), int(self.cfg.timesteps))
        loop = self._loop
        if loop is None or self._goals_key != key or self.cost._robot_model()[1] is not loop.robot:
            old = loop
            if old is not None and old._pending is not None:
                old.flush()
            loop = DeviceLoop(self.cost, self)
            if old is not None and old.state.shape == loop.state.shape:
                loop.state.copy_(old.state)
            elif self.__dict__.pop("_push_state", False):  # reset(): the host attributes are the truth, the device follows
                import torch
                hv, G = self.__dict__["_hv"], self.N
                vec = np.concatenate([np.asarray(hv["sum_costs"], np.float64), np.asarray(hv["p"], np.float64),
                                      np.concatenate([np.asarray(e, np.float64) for e in hv["experts_p"]]),
                                      np.asarray(hv["q"], np.float64), np.asarray(hv["experts_costs"], np.float64)])
                if vec.shape[0] == 7 * G + 10:
                    loop.state.copy_(torch.from_numpy(vec[None]))
            self._loop = self.cost._loop = loop
            self._goals_key = key
        elif self.cost.__dict__.get("_loop") is not loop:
            self.cost._loop = loop  # several learners may share a Cost: the optimiser follows the one that spoke last
        return loop

    def _params(self, alg: str, tiled: bool = False) -> _lib.LearnerParams:
        cfg = self.cfg
        p = _lib.LearnerParams()
        p.alg = _lib.ALG[alg]
        p.num_goals, p.n_waypoints = self.N, int(cfg.timesteps)
        p.start_idx = min(int((self.t / cfg.optim_steps) * cfg.timesteps), cfg.timesteps - 1)  # online_learner.py:109-110
        p.use_standoff = int(bool(cfg.use_standoff))
        p.constraint_num = int(self._loop.c)
        p.normalize_cost = int(bool(cfg.normalize_cost))
        p.base_obstacle_weight = float(cfg.base_obstacle_weight)
        p.smooth_weight = float(cfg.smoothness_base_weight * cfg.dist_eps)
        p.eta = float(self.eta)
        if tiled:  # the goal costs come from the latency-mode launch as [G][parts] partial sums
            p.cost_parts = ops.goalset_parts(p.n_waypoints - p.start_idx, self._loop.LAT_TILING[0])
        return p

    def _pull_state(self):
        G = self.N
        s = self._loop.pull_state()
        self.sum_costs, self.p = s[:G].copy(), s[G:2 * G].copy()
        self.experts_p = [s[2 * G + i * G: 3 * G + i * G].copy() for i in range(5)]
        self.q, self.experts_costs = s[7 * G:7 * G + 5].copy(), s[7 * G + 5:7 * G + 10].copy()
        if self.alg_name == "Proj":  # online_learner.py:200-210: one-hot on the goal closest to the trajectory's end
            self.p = np.zeros(self.N)
            self.p[int(self._loop._m_idx)] = 1

    def _goal_taken(self, idx: int):
        """Host bookkeeping of an update whose goal index has just become known (online_learner.py:243-248)."""
        if self.alg_name in ("FTL", "FTC"):
            self.last_leader = idx
        self.Ti[idx] += 1
        self.Tis.append(self.Ti)

    # ---- the reference's methods -------------------------------------------------------------------
    def cost_vector(self):
        """Objective cost estimate per goal at the current self.t (online_learner.py:104-160); the state is untouched."""
        cfg = self.cfg
        reach = self.env.objects[self.env.target_idx].reach_grasps
        if getattr(cfg, "traj_init", "grasp") == "grasp" and (len(reach) == 0 or (cfg.use_standoff and len(np.array(reach).shape) == 2)):
            return np.zeros(1)
        loop = self._tensors()
        if loop._pending is not None:
            loop.flush()
        loop.sync_inputs(self.traj)
        prm = self._params("FTC")  # FTC keeps no state: only the cost vector is of interest
        model = self.cost._robot_model()[0]
        traj = loop.d("traj")
        G = self.N
        gcost, gcol = loop.gcost.view(-1)[:G].view(1, G), loop.gcol.view(-1)[:G].view(1, G)
        ops.goalset_cost(loop.robot, model.points_per_link, self.cost._scenes(), traj[:, prm.start_idx], loop.cv_goals,
                         prm.n_waypoints - prm.start_idx, float(cfg.time_interval), soften_fingers=False, out=(gcost, gcol))
        ops.goal_update(prm, traj, loop.goal_set, loop.reach, gcost, loop.state.clone(), loop.s_idx, loop.s_end, loop.s_rows, loop.s_gp, loop.cv)
        return loop.cv[0].cpu().numpy()

    def update_goal_dist(self):
        """One step of the configured rule on the goal distribution (online_learner.py:162-234), at once."""
        loop = self._tensors()
        idx = loop.update_now(self._params(self.alg_name, tiled=True))
        if self.alg_name in ("FTL", "FTC"):
            self.last_leader = idx
        return idx

    def reset(self, traj):
        """Reset the online learner for a new trajectory (omg/online_learner.py:251-263): alg_name, N, T, weights, t, traj, p,
        sum_costs and last_leader start again; the experts' distributions, their costs, the mixture weights q and the Ti / Tis
        counters stay what they are, as in the reference.  (A goal set of another SIZE leaves the reference's experts_p at the
        old length, which its MD update cannot use; here the experts then start from uniform like in __init__.)"""
        old = self._loop
        if old is not None:
            if old._pending is not None:
                old.flush()
            if old.state_dirty:
                self._pull_state()  # experts_p / q / experts_costs of the device are the ones that stay
        n_old = self.N
        self.alg_name = self.cfg.ol_alg
        if self.alg_name not in _lib.ALG:
            raise ValueError(f"cfg.ol_alg = {self.alg_name!r}: the learner knows {sorted(_lib.ALG)}")
        self.N = n = len(traj.goal_set)
        self.T = self.cfg.optim_steps
        self.weights = np.ones(n)
        self.t = 0.0
        self.traj = traj
        self.p, self.sum_costs = np.ones(n) / n, np.zeros(n)
        self.last_leader = 0
        if n != n_old:
            self.experts_p = [np.ones(n) / n for _ in self.etas]
        # the device loop belongs to a (learner, trajectory) pair: the next use builds one for `traj` and uploads this state
        self._loop, self._goals_key = None, None
        if self.cost.__dict__.get("_loop") is old:
            self.cost._loop = None
        self.__dict__["_push_state"] = True

    def update_goal(self):
        """Take the arg-max of the goal distribution (online_learner.py:237-249); True when the goal changed.  With
        DEFER_UPDATE the device work waits for the Optimizer.optimize() that follows (or for whoever looks at the result
        first): traj.goal_idx / traj.end / the return value are promises that turn into int / array / bool on use."""
        self.t += 1
        loop = self._tensors()
        if self.DEFER_UPDATE:
            from .device_loop import _LazyEnd
            idx, changed = loop.defer_update(self._params(self.alg_name, tiled=True))
            self.traj.goal_idx = idx
            self.traj.end = _LazyEnd(loop, idx)
            return changed
        goal_idx_old = self.traj.goal_idx
        self.traj.goal_idx = self.update_goal_dist()  # np.argmax(self.p) with numpy's NaN / tie rules, taken on the device
        self.traj.end = self.traj.goal_set[self.traj.goal_idx]
        self.Ti[self.traj.goal_idx] += 1
        self.Tis.append(self.Ti)
        return self.traj.goal_idx != goal_idx_old
