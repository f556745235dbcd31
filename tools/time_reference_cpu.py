#!/usr/bin/env python3
"""Time the REFERENCE's own CPU path on bench.py's workload — build container only.

The reference Python (omg/cost.py, omg/optimizer.py, omg/online_learner.py, robot_pykdl.forward_kinematics_parallel) is
imported from /root/reference through the harness of tests/golden/make_golden.py (stubbed optional modules; `omg_cuda`
= the oracle's C restatement of the CUDA-only op, because layers/ cannot be built here).  It never travels to the GPU
box: the numbers this prints are copied into BASELINE.md by hand (section 3.1), with the core count.

    python tools/time_reference_cpu.py [--goals 64] [--iters 5]

Timed, on ONE scene of bench.py's workload (4 x 64^3 objects + a 128x96x32 table, 30 waypoints, 150 collision points):
  * Optimizer.optimize(traj, force_update=True)          omg/optimizer.py:115-135  (Cost.compute_total_loss + update)
  * Learner.update_goal()                                omg/online_learner.py:237-249 -> cost_vector -> Cost.batch_obstacle_cost
                                                         (omg/cost.py:192-286) over `goals` x 30 interpolated waypoints
  * one planner-loop iteration = update_goal + optimize  omg/planner.py:612-621  (what bench.py's "step" does per scene)
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import time
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.dont_write_bytecode = True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--op-threads", type=int, default=0, help="OpenMP threads of the stand-in SDF op (0 = all cores)")
    args = ap.parse_args()
    if not Path("/root/reference/omg/cost.py").exists():
        raise SystemExit("needs the reference tree (/root/reference): build container only")
    spec = importlib.util.spec_from_file_location("make_golden", ROOT / "tests" / "golden" / "make_golden.py")
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    config, cost_mod, opt_mod, util, rk = mg.load_reference()
    cfg = config.cfg
    from oracle import oracle as orc
    from omg_planner_amd import robot as rb, scenes as sc

    cores = os.cpu_count() or 1
    orc.set_threads(args.op_threads or cores)
    kin = mg.make_kinematics(rk)
    model = rb.PandaModel(seed=0)
    n, G = 30, args.goals
    mg.reset_cfg(cfg, timesteps=n, use_standoff=False, ol_alg="MD")  # bench.py: use_standoff False (omg.core -exp), MD = reference default
    scene = sc.make_tabletop_scene(0, grid=64)  # scene 0 of bench.py's workload
    env, sdf, lim = mg.make_env(cfg, kin, model, scene)
    goals = sc.make_reach_goals(scene, model, G, 0)
    env.objects[env.target_idx].reach_grasps = goals[:, None, :]
    start = rb.HOME_CONFIG.copy()
    c = cost_mod.Cost(env)
    traj = mg.Traj(cfg, sc.cubic_init(start, goals[0], n), start, goals[0], goal_set=goals, goal_idx=0)
    traj.interpolate_waypoints = lambda waypoints=None, mode="cubic": traj.set(
        util.interpolate_waypoints(np.stack([traj.start, traj.end]), cfg.timesteps, 9, mode=mode))
    learner = mg.LEARNER_MOD.Learner(env, traj, c)
    optim = opt_mod.Optimizer(types.SimpleNamespace(config=cfg, robot=env.robot), c)

    def timed(fn, reps):
        fn()  # warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), float(np.min(ts))

    opt_med, opt_min = timed(lambda: optim.optimize(traj, force_update=True), max(args.iters, 5))
    learner.t = 0  # keep the goal-set window at the full 30 waypoints like bench.py's step
    def goal_step():
        learner.t = 0
        learner.update_goal()
    ug_med, ug_min = timed(goal_step, args.iters)
    it = opt_med + ug_med
    out = {"host_cores": cores, "sdf_op": f"oracle/omg_oracle.c stand-in for omg_cuda, {args.op_threads or cores} OpenMP thread(s)",
           "scene": "bench.py scene 0 (4 x 64^3 + 128x96x32 table), 30 waypoints, 150 collision points", "goals": G,
           "optimize_ms_median": opt_med * 1e3, "optimize_ms_min": opt_min * 1e3,
           "update_goal_s_median": ug_med, "update_goal_s_min": ug_min,
           "planner_iteration_s": it, "scene_iterations_per_s": 1.0 / it,
           "plan_70_iterations_s_estimate": 50 * it + 20 * opt_med}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
