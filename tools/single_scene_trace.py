"""One scene x 64 goals: plan() a few times (run under `rocprofv3 --kernel-trace` and read the dispatches of the last plan with
tools/kernel_timeline.py), print the wall time per plan.  --latency: the engine's latency mode."""
import argparse
import copy
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd.engine import ChompEngine  # noqa: E402

import os as _os
if _os.environ.get("OMGX_LAT_PARTS"):
    ChompEngine.LAT_GOAL_PARTS = int(_os.environ["OMGX_LAT_PARTS"])  # experiments


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=1)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--plans", type=int, default=4)
    ap.add_argument("--latency", type=int, default=-1)
    ap.add_argument("--graph", action="store_true")
    args = ap.parse_args()
    cfg, model, batch, start, goals = bench.build_workload(args.scenes, args.goals, 30, 64, 0, False)
    kw = {}
    if args.latency >= 0:
        kw["latency_mode"] = bool(args.latency)
    for i in range(args.plans):
        e = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device="cuda:0", ol_alg="MD", **kw)
        if args.graph:
            fresh = e.snapshot()
            pg = e.capture_plan(early_stop=True)
            for _ in range(3):
                e.restore(fresh)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                pg.replay()
                torch.cuda.synchronize()
                print("graph plan %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3))
        else:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.plan(early_stop=False)
            torch.cuda.synchronize()
            print("plan %d: %.3f ms  cost %.9g" % (i, (time.perf_counter() - t0) * 1e3, float(e.final_costs()[0])))


if __name__ == "__main__":
    main()
