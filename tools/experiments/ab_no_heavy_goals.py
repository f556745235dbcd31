"""Experiment (round 4): is a mid-size batch's step set by its HEAVIEST goal workgroups?
Measures every goal workgroup's duration, replaces the goals that run longer than `--cap` x their scene's median by the scene's
median goal (same number of workgroups, no heavy tail) and times the bench step of both workloads in the layout rule's layout.
An upper bound for what splitting the heavy goals over several workgroups can return (a split repeats the prologue).
    python tools/ab_no_heavy_goals.py --scenes 13 --goals 128 [--cap 1.2]"""
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

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=13)
    ap.add_argument("--goals", type=int, default=128)
    ap.add_argument("--cap", type=float, default=1.2)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    from omg_planner_amd.engine import ChompEngine
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, 30, 64, 0, False)

    def timed(goal_set):
        eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goal_set, device=dev, ol_alg="MD")
        snap = eng.snapshot()
        for _ in range(10):
            eng.t = 0
            eng.iterate(0)
        eng.join()
        torch.cuda.synchronize()
        eng.restore(snap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            eng.t = 0
            eng.iterate(0)
        eng.join()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / a.iters * 1e3, 4)

    ChompEngine.MEASURE_MIN_ITEMS = 1
    one = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    for _ in range(4):
        one.t = 0
        one.iterate(0)
    torch.cuda.synchronize()
    assert one._measured
    work = one.work[: a.scenes * a.goals].cpu().numpy().astype(np.float64).reshape(a.scenes, a.goals)
    ChompEngine.MEASURE_MIN_ITEMS = 256
    light = goals.copy()
    replaced = 0
    for s in range(a.scenes):
        med = np.median(work[s])
        mid = int(np.argsort(work[s])[a.goals // 2])
        heavy = work[s] > a.cap * med
        light[s, heavy] = goals[s, mid]
        replaced += int(heavy.sum())
    res = {"scenes": a.scenes, "goals": a.goals, "cap": a.cap, "workgroup_us_mean/p90/max": [round(float(x) / 100, 1) for x in (work.mean(), np.percentile(work, 90), work.max())],
           "goals_replaced": replaced, "ms_per_step": timed(goals), "ms_per_step_without_heavy_goals": timed(light)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
