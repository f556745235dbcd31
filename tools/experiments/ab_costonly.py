#!/usr/bin/env python3
"""A/B timing of the COST-ONLY goal-set launch (omgx_goalset_cost: goal workgroups, no trajectory layer, scene-major order) for
library variants — isolates the goal path of k_goalset_queue.  One process per variant:
    python tools/ab_costonly.py --lib X.so --waypoints 22
Prints one JSON line (median / min / mean launch duration from HIP events attached to every dispatch)."""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--waypoints", type=int, default=30)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    if a.lib:
        _lib.LIB_PATH = Path(a.lib).resolve()
    from omg_planner_amd import ops
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, a.waypoints, 64, 0, False)
    eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD")
    lib = _lib.lib()
    n = a.waypoints

    def launch():
        ops.goalset_cost(eng.robot, eng.P, eng.scenes, eng.traj[:, 0], eng.cv_goals, n, cfg.time_interval, out=(eng.goal_cost, eng.goal_col))

    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    lib.omgx_timing_enable(1)
    for _ in range(a.iters):
        launch()
    torch.cuda.synchronize()
    buf, kinds = (C.c_float * 4096)(), (C.c_int32 * 4096)()
    k = lib.omgx_timing_collect(buf, kinds, 4096)
    lib.omgx_timing_enable(0)
    d = np.array([buf[i] for i in range(k)]) * 1e3
    print(json.dumps({"tag": a.tag, "waypoints": n, "costonly_us_median": round(float(np.median(d)), 1), "min": round(float(d.min()), 1),
                      "mean": round(float(d.mean()), 1), "launches": int(k), "checksum": float(eng.goal_cost.double().sum().item())}))


if __name__ == "__main__":
    main()
