"""Shared helpers for the parity tests: fixture loading and scene-table reconstruction."""
from __future__ import annotations

from pathlib import Path

import numpy as np

from omg_planner_amd import robot as rb
from omg_planner_amd import scenes as sc

GOLDEN = Path(__file__).resolve().parent / "golden"

# reference defaults of the SDF layer parameters (omg/config.py:42-53)
LAYER_CFG = dict(epsilon=0.2, target_epsilon=0.1, clearance=0.01, target_clearance=0.0)


def load(name: str) -> dict:
    with np.load(GOLDEN / name, allow_pickle=False) as d:
        return {k: d[k] for k in d.files}


def model_from(fx: dict) -> rb.PandaModel:
    return rb.PandaModel(collision_points=fx["collision_points"])


def layer_params_from(fx: dict):
    """Per-object parameters exactly as Cost.compute_obstacle_cost_layer builds them (cost.py:303-328)
    from the fixture's object names / poses / attached flags."""
    names = [str(n) for n in fx["obj_names"]]
    objs = [sc.SceneObject(n, fx["obj_pose"][i], None, bool(fx["attached"][i])) for i, n in enumerate(names)]
    scene = sc.Scene(objs, int(fx["target_idx"]))
    return sc.layer_params(scene, **LAYER_CFG)


def batch_from(fx: dict) -> sc.SceneBatch:
    """Single-scene engine table addressing the fixture's padded [O,X,Y,Z] tensor in place."""
    poses, eps, pad, clr, dis = layer_params_from(fx)
    table = sc.table_from_padded(poses, fx["limits"], eps, pad, clr, dis)
    return sc.SceneBatch(table, np.array([0, len(table)], np.int32), np.ascontiguousarray(fx["sdf"], np.float32).ravel())


def params_from(fx: dict, params_cls, n: int, P: int, do_update: int, obstacle_weight: float, smoothness_weight: float,
                step_size: float = 0.1, reach_tail_length: int = 5):
    """omgx_chomp_params from a fixture's recorded cfg_* scalars + reference defaults (omg/config.py)."""
    p = params_cls()
    p.n_waypoints, p.n_points = n, P
    p.top_k = int(fx["cfg_top_k"])
    p.consider_finger = int(fx.get("cfg_consider_finger", 0))
    p.goal_set_proj = int(fx["cfg_goal_set_proj"])
    p.use_standoff = int(fx["cfg_use_standoff"])
    p.constraint_num = reach_tail_length if p.use_standoff else 1
    p.uncheck_finger_collision = int(fx.get("cfg_uncheck", 0))
    p.joint_limit_max_steps = int(fx.get("cfg_joint_limit_max_steps", 10))
    p.allow_collision_point = int(fx.get("cfg_allow_collision_point", 5))
    p.pre_terminate = int(fx.get("cfg_pre_terminate", 1))
    p.do_update = do_update
    p.time_interval = float(fx["cfg_dt"])
    p.obstacle_weight = obstacle_weight
    p.smoothness_weight = smoothness_weight
    p.step_size = step_size
    p.clip_grad_scale = float(fx.get("cfg_clip_grad_scale", 10.0))
    p.terminate_smooth_loss = float(fx.get("cfg_terminate_smooth_loss", 35.0))
    for d in range(9):
        p.link_smooth_weight[d] = 1.0
    return p


INFO_IDX = {k: i for i, k in enumerate(["cost", "obs", "smooth", "weighted_obs", "weighted_smooth", "weighted_obs_grad",
                                        "weighted_smooth_grad", "grad", "collide", "reach", "terminate",
                                        "failure_terminate", "execute", "standoff_idx", "violate_limit", "limit_steps"])}
