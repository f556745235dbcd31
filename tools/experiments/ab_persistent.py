"""A/B: the bench step launched per iteration (pipelined parts) against the persistent launch (omgx_plan_persistent), same workload.
    python tools/experiments/ab_persistent.py [--scenes 100 --goals 64 --waypoints 30 --objects 4 --steps 50 --regions 6]"""
import argparse
import copy
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import os  # noqa: E402

from omg_planner_amd import _lib  # noqa: E402

if os.environ.get("OMGX_LIB"):  # a variant build of the library (omg-planner_amd/csrc/<name>)
    _lib.LIB_PATH = ROOT / "omg-planner_amd" / "csrc" / os.environ["OMGX_LIB"]
import bench  # noqa: E402
from omg_planner_amd.engine import ChompEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--waypoints", type=int, default=30)
    ap.add_argument("--objects", type=int, default=4)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--regions", type=int, default=6)
    ap.add_argument("--alg", default="MD")
    ap.add_argument("--max-wg", type=int, default=0)
    ap.add_argument("--update-cus", type=int, default=-1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(args.scenes, args.goals, args.waypoints, args.grid, 0, False, num_objects=args.objects, device=dev)
    out = {"shape": [args.scenes, args.goals, args.waypoints, args.objects], "steps": args.steps, "lib": Path(_lib.LIB_PATH).name, "update_cus": args.update_cus}
    engs = {}
    for mode in ("launches", "persistent"):
        eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=args.alg)
        eng.pose_hand_over(True)
        snap = eng.snapshot()
        engs[mode] = (eng, snap)

    def region(mode):
        eng, snap = engs[mode]
        eng.restore(snap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "launches":
            for _ in range(args.steps):
                eng.t = 0
                eng.iterate(0)
            eng.join()
        else:
            eng.run_persistent([0] * args.steps, pin_window=True, max_workgroups=args.max_wg, update_cus=args.update_cus)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    for mode in engs:
        region(mode)  # warm
    res = {m: [] for m in engs}
    for _ in range(args.regions):
        for m in engs:
            res[m].append(region(m))
    for m in engs:
        out[m + "_ms_per_step"] = float(np.median(res[m]))
        out[m + "_spread"] = [float(min(res[m])), float(max(res[m]))]
    a, b = engs["launches"][0], engs["persistent"][0]
    out["bits_equal"] = {k: bool(torch.equal(getattr(a, k), getattr(b, k))) for k in ("traj", "info", "goal_idx", "learner_state", "goal_cost")}
    out["status"] = b.persistent_status()
    out["layout"] = getattr(a, "layout_used", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
