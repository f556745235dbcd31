"""A/B timing of the goal-set kernel on the bench workload (GPU box).  One process per variant, run them alternately:
    for i in 1 2 3; do python tools/ab_goalset.py --lib A.so; python tools/ab_goalset.py --lib B.so; done
Prints one JSON line: median / min / mean duration of the goal-set kernel (HIP events attached to every dispatch) and of the
whole iteration (wall clock over the timed block)."""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None, help="library variant (default: the shipped libomg_hip.so)")
    ap.add_argument("--sched", default="auto", choices=["none", "auto"])
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--waypoints", type=int, default=30)
    ap.add_argument("--objects", type=int, default=4, help="objects per scene besides the table")
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = Path(args.lib).resolve()
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(args.scenes, args.goals, args.waypoints, 64, 0, False, num_objects=args.objects)
    eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD")
    eng.auto_schedule = args.sched == "auto"
    lib = _lib.lib()
    snap = eng.snapshot()

    def block(iters):
        eng.restore(snap)
        for _ in range(iters):
            eng.t = 0
            eng.iterate(0)

    block(10)
    torch.cuda.synchronize()
    lib.omgx_timing_enable(1)
    t0 = time.perf_counter()
    block(args.iters)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.iters
    buf = (C.c_float * 4096)()
    kinds = (C.c_int32 * 4096)()
    n = lib.omgx_timing_collect(buf, kinds, 4096)
    lib.omgx_timing_enable(0)
    d = np.array([buf[i] for i in range(n) if kinds[i] == 0]) * 1e3
    print(json.dumps({"tag": args.tag or (Path(args.lib).name if args.lib else "shipped"), "sched": args.sched, "goalset_us_median": round(float(np.median(d)), 1),
                      "min": round(float(d.min()), 1), "mean": round(float(d.mean()), 1), "launches": int(len(d)), "iteration_us_wall": round(wall * 1e6, 1)}))


if __name__ == "__main__":
    main()
