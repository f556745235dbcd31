#!/usr/bin/env python3
"""Whole-plan wall time of the bench workload under the engine's scheduling policies (needs a GPU).

    python tools/ab_plan.py [--scenes 100] [--goals 64] [--reps 3]

Prints, per policy, the best-of-`reps` time of ChompEngine.plan with and without early stop:
  none      scene-major order (the kernel deals scenes to XCDs itself)
  sched/0   measured schedule; under early stop back to scene-major once scenes drop out
  sched/k   measured schedule, rebuilt without the terminated scenes every k iterations (omgx_goalset_schedule)
"""
import argparse
import copy
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch
    import bench
    from omg_planner_amd.engine import ChompEngine
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, 30, a.grid, 0, False)

    def run(auto, every, early, pipeline=1):
        best, host = float("inf"), float("inf")
        for _ in range(a.reps):
            eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
            eng.auto_schedule, eng.reschedule_every, eng.pipeline = auto, every, pipeline
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.plan(early_stop=early)
            host = min(host, (time.perf_counter() - t0) * 1e3)  # when the host has enqueued the whole plan
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best, int((eng.active == 0).sum().item()), host

    run(True, 1, True)  # warm-up: code objects, allocator
    for name, auto, every in (("none", False, 0), ("sched/0", True, 0), ("sched/1", True, 1), ("sched/2", True, 2), ("sched/4", True, 4),
                              ("sched/8", True, 8)):
        full, _, _ = run(auto, every, False)
        early, term, _ = run(auto, every, True)
        print(f"{name:8s} plan {full:7.2f} ms   early-stop {early:7.2f} ms   ({term} of {a.scenes} scenes terminated)", flush=True)
    for k in (1, 2):
        full, _, h1 = run(True, 0, False, k)
        early, term, h2 = run(True, 0, True, k)
        print(f"pipeline {k}: plan {full:7.2f} ms (host enqueue {h1:6.2f})   early-stop {early:7.2f} ms (host enqueue {h2:6.2f})", flush=True)


if __name__ == "__main__":
    main()
