#!/usr/bin/env python3
"""Experiment: the pipeline's update launches on a HIGH-priority stream of their own, the goal-set launches on normal streams.

The kernel trace of the two-part pipeline (tools/kernel_timeline.py) shows two regimes: both parts' goal-set launches running
concurrently and then both update launches (290 us per iteration of all scenes), or — when the parts drift apart — an update launch
STARVED for the whole duration of the other part's goal-set launch (its 94 KB workgroups never find a CU while 32 KB goal-set
workgroups keep refilling every slot that frees): 338 us.  Would the update launch get its CUs if its queue had priority?

    python tools/ab_prio_update.py --parts 2|3 [--prio 0|-1] [--scenes 100]
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--prio", type=int, default=-1, help="priority of the update streams (-1 high, 0 normal)")
    ap.add_argument("--shared", action="store_true", help="one update stream for all parts")
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    from omg_planner_amd.engine import ChompEngine
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, 30, 64, 0, False)
    cuts = [a.scenes * k // a.parts for k in range(a.parts + 1)]
    parts = []
    shared_up = torch.cuda.Stream(device=dev, priority=a.prio)
    for k in range(a.parts):
        lo, hi = cuts[k], cuts[k + 1]
        st_gs = torch.cuda.Stream(device=dev)
        st_up = shared_up if a.shared else torch.cuda.Stream(device=dev, priority=a.prio)
        with torch.cuda.stream(st_gs):
            e = ChompEngine(model, batch.subset(lo, hi), cfg, start[lo:hi], goals[lo:hi], device=dev, ol_alg="MD")
            for t in range(4):  # schedule measured, prepared calls built
                e.t = 0
                e.iterate(0)
        parts.append((e, st_gs, st_up, torch.cuda.Event(), torch.cuda.Event()))
    torch.cuda.synchronize()
    for e, st_gs, st_up, ev_gs, ev_up in parts:
        ev_up.record(st_up)

    def step():
        for e, st_gs, st_up, ev_gs, ev_up in parts:
            calls = e._hot[1]
            st_gs.wait_event(ev_up)
            e.t = 1
            prm = e._learner_params()
            calls.goalset_layer(prm.start_idx, False, e.schedule, None, st_gs.cuda_stream)
            ev_gs.record(st_gs)
            st_up.wait_event(ev_gs)
            e._schedule()
            e._ticket += 1
            calls.update(prm, e._params(True), 2 * e.S <= e._num_cus, e._ticket, False, st_up.cuda_stream)
            ev_up.record(st_up)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"parts": a.parts, "update_stream_priority": a.prio, "shared": a.shared, "us_per_iteration_of_all_scenes": round(dt / a.iters * 1e6, 1),
                      "scene_iterations_per_s": round(a.scenes * a.iters / dt)}))


if __name__ == "__main__":
    main()
