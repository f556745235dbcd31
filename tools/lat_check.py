"""Latency mode against the batch layout on the same scenes (GPU box): layer outputs bit for bit, goal costs within float32
summation rounding, whole plans within the free-running tolerance; wall time per plan in both modes."""
import copy
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd.engine import ChompEngine  # noqa: E402

import os as _os
if _os.environ.get("OMGX_LAT_NO_POSES"):
    ChompEngine.LAT_HAND_OVER_POSES = False
if _os.environ.get("OMGX_LAT_PARTS"):
    ChompEngine.LAT_GOAL_PARTS = int(_os.environ["OMGX_LAT_PARTS"])  # experiments


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 64, 0, False)
    mk = lambda lat: ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device="cuda:0", ol_alg="MD", latency_mode=lat)
    a, b = mk(False), mk(True)
    for e in (a, b):
        e.select_initial_goal()
    print("initial goal equal:", torch.equal(a.goal_idx, b.goal_idx), "traj equal:", torch.equal(a.traj, b.traj))
    for t in (0, 7, 23, 49):
        for e in (a, b):
            e.t = t
            e.iterate(t)
        torch.cuda.synchronize()
        ga, gb = a.goal_cost_total().double(), b.goal_cost_total().double()
        rel = ((ga - gb).abs() / ga.abs().clamp_min(1e-6)).max().item()
        print(f"t={t}: parts {b._parts_last} goal cost max rel diff {rel:.3e}; layer pot equal {torch.equal(a.pot, b.pot)} grad equal {torch.equal(a.pgrad, b.pgrad)} col equal {torch.equal(a.col, b.col)}; "
              f"goal_idx equal {torch.equal(a.goal_idx, b.goal_idx)}; traj max diff {(a.traj - b.traj).abs().max().item():.3e}")
    res = {}
    for lat in (False, True):
        best = 1e9
        for _ in range(4):
            e = mk(lat)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            info = e.plan(early_stop=False)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        res[lat] = (best, e.traj.clone(), info.clone())
        fresh = e.snapshot()
        e2 = mk(lat)
        pg = e2.capture_plan(early_stop=True)
        fr = e2.snapshot()
        gb_ = 1e9
        for _ in range(4):
            e2.restore(fr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pg.replay()
            torch.cuda.synchronize()
            gb_ = min(gb_, (time.perf_counter() - t0) * 1e3)
        print(f"latency_mode={lat}: plan {best:.3f} ms eager, {gb_:.3f} ms as one graph")
    print("plans: traj max diff %.3e, cost rel diff %.3e" % ((res[False][1] - res[True][1]).abs().max().item(),
          ((res[False][2][:, 0] - res[True][2][:, 0]).abs() / res[False][2][:, 0].abs()).max().item()))


if __name__ == "__main__":
    main()
