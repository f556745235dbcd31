"""Planner.setup_goal_set (omg/planner.py:502-597): prune a target's goal set before planning.

Device part: the collision filter's `Cost.batch_obstacle_cost(goal_set, special_check_id=i,
uncheck_finger_collision=-1)` for S scenes at once = one `omgx_fk_sdf(soften_fingers=1)` over [S, G0] goal configurations,
reduced per goal.  Host part: the thresholding, the greedy diversity filter and the sampling, restated with the
reference's indexing quirks so that the same `np.random` stream picks the same goals.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def goal_collision_stats(robot, P: int, scenes: "ops.DeviceScenes", goal_sets: torch.Tensor):
    """goal_sets [S,G0,9] float64 (device) -> (collide [S,G0], potentials [S,G0]) float32: per-goal number of colliding
    (link, point, object) lookups and summed potentials with the finger links softened (x0.1, collisions ignored,
    cost.py:350-353), as planner.py:512-524 reduces them."""
    pot, _, col = ops.fk_sdf(robot, P, scenes, goal_sets, soften_fingers=True)
    return col.sum(dim=(-2, -1)), pot.sum(dim=(-2, -1))


def select_goals(goal_set, reach_goal_set, collide, potentials, allow_collision_point: int = 5, goal_set_max_num: int = 100,
                 filter_collision: bool = True, filter_diversity: bool = True, rng=np.random):
    """The host logic of planner.py:526-575 for ONE target object.  Returns (grasps, reach_grasps, potentials, chosen)
    where `chosen` indexes the collision-filtered list — or ([], [], [], []) when nothing survives ("IK FAIL").

    Quirks kept on purpose (they decide which goals the reference ends up with):
      * the diversity filter walks goal_set[1:] but records the loop counter j, i.e. the index of the PREVIOUS element;
        goal 0 seeds `unique_grasps` yet only enters `indexes` through that off-by-one (planner.py:548-558);
      * a candidate closer than 0.5 (joint-space L2) to any kept goal is dropped;
      * `np.random.choice(indexes, min(num, goal_set_max_num), replace=False)` draws from the global numpy stream."""
    goal_set = [np.asarray(g) for g in goal_set]
    reach_goal_set = list(reach_goal_set) if reach_goal_set is not None else []
    collide = np.asarray(collide)
    potentials = np.asarray(potentials)
    if filter_collision:
        free = (collide <= allow_collision_point).nonzero()[0]
        goal_set = [goal_set[i] for i in free]
        try:
            reach_goal_set = [reach_goal_set[i] for i in free]
        except Exception:  # noqa: BLE001  (planner.py:533-536: a short reach list is silently kept as it is)
            pass
        potentials = potentials[free]
    num = len(goal_set)
    indexes = list(range(num))
    if filter_diversity and num > 0:
        unique = [goal_set[0]]
        indexes = []
        for j, joint in enumerate(goal_set[1:]):
            if np.amin(np.linalg.norm(np.array(unique) - joint, axis=-1)) < 0.5:
                continue
            unique.append(joint)
            indexes.append(j)  # sic: j, not j + 1
        num = len(indexes)
    if num == 0:
        return [], [], [], []
    chosen = rng.choice(indexes, min(num, goal_set_max_num), replace=False)
    grasps = [goal_set[int(i)] for i in chosen]
    reach = np.array([reach_goal_set[int(i)] for i in chosen]) if reach_goal_set else np.zeros((0,))
    return grasps, reach, potentials[chosen], chosen
