"""Hyper-parameters of the CHOMP path: the subset of the reference's global ``cfg`` (omg/config.py:27-131)
that the hot path reads, with the same names and defaults, as a plain attribute namespace.

The reference builds dense finite-difference matrices, ``A = D^T D`` and ``Ainv = inv(A)`` at import time
(and needs a CUDA device to import, config.py:222-227,248).  The device kernels use closed forms instead
(csrc/omg_chomp.hip), so nothing here touches a GPU; ``diff_matrices`` / ``A`` / ``Ainv`` are still
provided (numpy, built on demand) for callers that read them.
"""
from __future__ import annotations

import numpy as np


class Config:
    def __init__(self, **overrides):
        # -- hyperparameters (config.py:30-39)
        self.smoothness_base_weight = 0.1
        self.base_obstacle_weight = 1.0
        self.base_grasp_weight = 1.0
        self.cost_schedule_decay = 1
        self.cost_schedule_boost = 1.02
        self.base_step_size = 0.1
        self.step_decay_rate = 1.0
        self.joint_limit_max_steps = 10
        self.optim_steps = 50
        # -- planner parameters (config.py:42-104)
        self.epsilon = 0.2
        self.target_epsilon = 0.1
        self.collision_point_num = 15
        self.time_interval = 0.1
        self.top_k_collision = 1000
        self.link_smooth_weight = np.ones(9)
        self.clearance = 0.01
        self.target_clearance = 0.0
        self.terminate_smooth_loss = 35
        self.goal_set_proj = True
        self.goal_set_max_num = 100
        self.ol_alg = "MD"
        self.dist_eps = 0.1
        self.goal_idx = -2
        self.pre_terminate = True
        self.uncheck_finger_collision = 0
        self.allow_collision_point = 5
        self.soft_joint_limit_padding = 0.2
        self.extra_smooth_steps = 20
        self.clip_grad_scale = 10.0
        self.normalize_cost = True
        self.disable_collision_set = []
        self.use_standoff = True
        self.consider_finger = False
        self.reach_tail_length = 5
        self.timesteps = 30
        self.report_cost = False
        self.report_time = False
        self.timeout = 3.0
        self.traj_interpolate = "cubic"   # config.py:63
        self.dynamic_timestep = False     # config.py:89: trajectory length from the start-goal distance (core.py:64-76)
        self.traj_delta = 0.05            # config.py:96
        self.traj_max_step = 50           # config.py:98
        self.traj_min_step = 2            # config.py:99
        self.base_link = "panda_link0"
        # -- scheduled by Optimizer.update (optimizer.py:68-80)
        self.obstacle_weight = self.base_obstacle_weight
        self.smoothness_weight = self.smoothness_base_weight
        self.grasp_weight = self.base_grasp_weight
        self.step_size = self.base_step_size
        for k, v in overrides.items():
            setattr(self, k, v)
        self.get_global_param(self.timesteps)

    # config.py:199-227.  NB the reference derives dt from the PREVIOUS timesteps value:
    # dt = 0.1 * cfg.timesteps / steps, then cfg.timesteps = steps.
    def get_global_param(self, steps=None):
        steps = self.timesteps if steps is None else steps
        self.time_interval = (0.1 * self.timesteps) / steps
        self.timesteps = steps
        self.diff_rule_length = 7
        self.diff_rule = np.array([[0, 0, -1, 1, 0, 0, 0], [0, 0, 1, -2, 1, 0, 0], [0, -0.5, 1, 0, -1, 0.5, 0]])
        self._mats = None

    def _build(self):
        if self._mats is None:
            n, dt = self.timesteps, self.time_interval
            mats = []
            for order, rule in enumerate(self.diff_rule, start=1):
                D = np.zeros((n + 1, n))
                for i in range(n + 1):
                    for j in range(-3, 3):
                        if 0 <= i + j < n:
                            D[i, i + j] = rule[j + 3]
                if self.goal_set_proj:
                    D[-1, -1] = 0
                mats.append(D / dt ** order)
            A = mats[0].T @ mats[0]
            self._mats = (mats, A, np.linalg.inv(A))
        return self._mats

    @property
    def diff_matrices(self):
        return self._build()[0]

    @property
    def A(self):
        return self._build()[1]

    @property
    def Ainv(self):
        return self._build()[2]

    # config.py:134-187: finite differences along the waypoint axis with the start/end boundary terms folded in
    def get_derivative(self, data, start, end, diff_rule=1):
        """data [..., n, 3] -> D_k data with x_{-1} = start, x_n = end; v_i = (x_i - x_{i-1})/dt, a_i = (x_{i-1} - 2x_i + x_{i+1})/dt^2."""
        n = data.shape[-2]
        D = self.diff_matrices[diff_rule - 1][: n + 1, :n]
        out = np.matmul(D, data)
        mid = self.diff_rule_length // 2
        scale = self.time_interval ** diff_rule
        out[..., 0, :] += self.diff_rule[diff_rule - 1][mid - 1] * start / scale
        out[..., -2, :] += self.diff_rule[diff_rule - 1][mid + 1] * end / scale
        out[..., -1, :] += self.diff_rule[diff_rule - 1][mid] * end / scale
        return out[..., :-1, :]

    def get_derivative_torch(self, data, start, end, diff_rule=1):
        """float32 torch twin of get_derivative (config.py:162-187)."""
        import torch
        n = data.shape[-2]
        D = torch.as_tensor(self.diff_matrices[diff_rule - 1][: n + 1, :n], dtype=torch.float32, device=data.device)
        out = torch.matmul(D, data)
        mid = self.diff_rule_length // 2
        scale = self.time_interval ** diff_rule
        out[..., 0, :] += float(self.diff_rule[diff_rule - 1][mid - 1]) * start / scale
        out[..., -2, :] += float(self.diff_rule[diff_rule - 1][mid + 1]) * end / scale
        out[..., -1, :] += float(self.diff_rule[diff_rule - 1][mid]) * end / scale
        return out[..., :-1, :]

    def layer_kwargs(self) -> dict:
        return dict(epsilon=self.epsilon, target_epsilon=self.target_epsilon, clearance=self.clearance,
                    target_clearance=self.target_clearance, disable_collision_set=tuple(self.disable_collision_set))


cfg = Config()
