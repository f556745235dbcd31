"""`Trajectory` — the data contract of omg/core.py:23-78: n interior waypoints x 9 dof between a
fixed start and a (goal-set selected) end.  Host-side container for the single-trajectory drop-in classes; the
batched engine keeps the same fields as device tensors."""
from __future__ import annotations

import numpy as np

from .config import cfg as _default_cfg
from .scenes import cubic_init, linear_init


class Trajectory(object):
    def __init__(self, timesteps=None, dof=9, cfg=None):
        self.cfg = cfg if cfg is not None else _default_cfg
        self.timesteps = self.cfg.timesteps if timesteps is None else timesteps
        self.dof = dof
        self.data = np.zeros([self.timesteps, dof])
        self.goal_set = []
        self.goal_quality = []
        self.goal_idx = 0
        self.start = np.array([0.0, -1.285, 0, -2.356, 0.0, 1.571, 0.785, 0.04, 0.04])
        self.end = np.array([-0.99, -1.74, -0.61, -3.04, 0.88, 1.21, -1.12, 0.04, 0.04])
        self.interpolate_waypoints(mode=getattr(self.cfg, "traj_interpolate", "cubic"))

    def update(self, grad):
        """data += grad on the arm joints (fingers only with cfg.consider_finger), fingers clamped to [0, 0.04]."""
        if self.cfg.consider_finger:
            self.data += grad
        else:
            self.data[:, :-2] += grad[:, :-2]
        self.data[:, -2:] = np.clip(self.data[:, -2:], 0, 0.04)

    def set(self, new_traj):
        self.data = new_traj

    def interpolate_waypoints(self, waypoints=None, mode="cubic"):
        """Interior waypoints linspace(0,1,n+2)[1:-1] between start and end; "cubic" is the clamped spline through
        the two knots (zero end slopes = the 3t^2 - 2t^3 blend), "linear" the straight line (util.py:238-258)."""
        cfg = self.cfg
        n = cfg.timesteps
        if getattr(cfg, "dynamic_timestep", False):  # core.py:64-76: length from the start-end distance; rebuilds dt and the matrices
            n = min(max(int(np.linalg.norm(self.start - self.end) / cfg.traj_delta), cfg.traj_min_step), cfg.traj_max_step)
            cfg.timesteps = n
            cfg.get_global_param(n)
        self.timesteps = n
        self.data = (cubic_init if mode == "cubic" else linear_init)(self.start, self.end, n)
