"""Oracle-driven planner iterations beside a running ChompEngine.  TEST INFRASTRUCTURE ONLY (like everything under
oracle/): used by tests/, __graft_entry__.smoke() and bench.py's `parity_sample` — as the checker of the timed workload,
never as the thing measured.

engine_vs_oracle() advances the engine by `steps` planner iterations (omg/planner.py:612-621: Learner.update_goal, then
Optimizer.optimize) and, for a few sample scenes, runs the same iterations through the CPU restatement
(oracle/omg_oracle.c: goalset_cost -> goal_update -> fk_sdf -> chomp_optimize) from the engine's own starting state and with
the parameter structs the engine used.  Returns the worst differences.
"""
from __future__ import annotations

import numpy as np

from . import oracle as orc


def _copy_struct(dst, src):
    for f, _ in dst._fields_:
        setattr(dst, f, getattr(src, f))
    return dst


def engine_vs_oracle(eng, batch, scene_ids, steps: int = 3, pin_window: bool = True, sophus: bool = False) -> dict:
    """(see _engine_vs_oracle)  sophus: the oracle's SDF pairs go through Sophus' quaternion round trip like the reference's kernel
    (omg_oracle.c, SOPHUS MODE) — the one place where the product path knowingly differs from the reference's arithmetic."""
    if not sophus:
        return _engine_vs_oracle(eng, batch, scene_ids, steps, pin_window)
    orc.set_sophus_mode(True)
    try:
        return dict(_engine_vs_oracle(eng, batch, scene_ids, steps, pin_window), sophus_mode=True)
    finally:
        orc.set_sophus_mode(False)


def _engine_vs_oracle(eng, batch, scene_ids, steps: int = 3, pin_window: bool = True) -> dict:
    """eng: ChompEngine (no early stop, no ragged goal sets); batch: the host SceneBatch it was built from.
    pin_window: keep Learner.t at 0 before every iteration like bench.py's step (goal-set window = all waypoints)."""
    import torch

    model, cfg = eng.model, eng.cfg
    blob, P, n, G = model.blob(), eng.P, eng.n, eng.G
    ids = [int(s) for s in scene_ids]
    torch.cuda.synchronize(eng.device)
    traj = {s: eng.traj[s].cpu().numpy().copy()[None] for s in ids}
    state = {s: eng.learner_state[s].cpu().numpy().copy()[None] for s in ids}
    start = {s: eng.start[s].cpu().numpy()[None] for s in ids}
    goals = {s: eng.goal_set[s].cpu().numpy()[None] for s in ids}
    reach = {s: (eng.reach[s].cpu().numpy()[None] if eng.reach is not None else None) for s in ids}
    cv_goals = {s: eng.cv_goals[s].cpu().numpy()[None] for s in ids}
    sub = {s: batch.subset(s, s + 1) for s in ids}
    end = {s: eng.end[s].cpu().numpy()[None] for s in ids}
    rows = {s: eng.goal_rows[s].cpu().numpy()[None] for s in ids}
    gpt = {s: eng.goal_point[s].cpu().numpy()[None] for s in ids}
    idx = {s: None for s in ids}
    info = {}
    for k in range(steps):
        if pin_window:
            eng.t = 0
        t_iter = 0 if pin_window else k
        select = cfg.goal_set_proj and t_iter < cfg.optim_steps and eng.ol_alg not in ("Baseline", "Proj")
        eng.iterate(t_iter)
        lp = _copy_struct(orc.LearnerParams(), eng._learner_params()) if select else None  # eng.t was advanced by the iteration
        cp = _copy_struct(orc.ChompParams(), eng._params(True))                             # after Optimizer.update's schedule
        for s in ids:
            if select:
                gc, _ = orc.goalset_cost(blob, P, sub[s], traj[s][:, lp.start_idx], cv_goals[s], n - lp.start_idx, cfg.time_interval)
                i_, end[s], rows[s], gpt[s], _ = orc.goal_update(lp, traj[s], goals[s], reach[s], gc, state[s])
                idx[s] = int(i_[0])
            pot, pg, col = orc.fk_sdf(blob, P, sub[s], traj[s], soften_fingers=cfg.uncheck_finger_collision == -1)
            traj[s], _, _, info[s] = orc.chomp_optimize(blob, cp, traj[s], start[s], end[s], rows[s], gpt[s], pot, pg, col)
    torch.cuda.synchronize(eng.device)
    d_traj = eng.traj.cpu().numpy()
    d_info = eng.info.cpu().numpy()
    d_idx = eng.goal_idx.cpu().numpy()
    out = {"scenes": ids, "steps": steps, "max_traj_err": 0.0, "max_cost_rel_err": 0.0, "goal_idx_equal": True}
    for s in ids:
        out["max_traj_err"] = max(out["max_traj_err"], float(np.abs(d_traj[s] - traj[s][0]).max()))
        ref_cost = float(info[s][0, 0])
        out["max_cost_rel_err"] = max(out["max_cost_rel_err"], abs(float(d_info[s, 0]) - ref_cost) / max(abs(ref_cost), 1e-12))
        if idx[s] is not None and int(d_idx[s]) != idx[s]:
            out["goal_idx_equal"] = False
    # north_star's bar: trajectory states and cost values within 1e-4
    out["ok"] = bool(out["goal_idx_equal"] and out["max_traj_err"] <= 1e-4 and out["max_cost_rel_err"] <= 1e-4)
    return out
