#!/usr/bin/env python3
"""Experiment (round 4, verdict item 1): what bounds the step of a small / mid-size batch — the host's launch rate or the GPU?

For a shape (scenes x goals) and a number of pipeline parts k = 1..K: ms per step of bench.py's step (window pinned at the
full n waypoints) enqueued eagerly, and the same steps captured once into a HIP graph and replayed (no host in the loop).

    python tools/ab_parts_graph.py --scenes 13 --goals 128 [--parts 1,2,3,4,6] [--iters 100] [--latency]
Prints one JSON line per k.
"""
import argparse
import copy
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
    ap.add_argument("--scenes", type=int, default=13)
    ap.add_argument("--goals", type=int, default=128)
    ap.add_argument("--parts", default="1,2,3,4,6")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--block", type=int, default=10, help="iterations per captured graph")
    ap.add_argument("--latency", action="store_true")
    ap.add_argument("--split", default="auto", help="auto | 0 | 1: learner and step in two workgroups")
    ap.add_argument("--goal-parts", type=int, default=1, help="workgroups per goal in the batch layout (ChompEngine(goal_parts=...))")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--measure-min", type=int, default=None, help="ChompEngine.MEASURE_MIN_ITEMS for this run")
    a = ap.parse_args()
    from omg_planner_amd.engine import ChompEngine
    if a.measure_min is not None:
        ChompEngine.MEASURE_MIN_ITEMS = a.measure_min
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, 30, 64, 0, False)
    for k in [int(x) for x in a.parts.split(",")]:
        if k > a.scenes or (a.latency and k > 1):
            continue
        eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", latency_mode=a.latency, goal_parts=a.goal_parts)
        eng.pipeline = k
        if a.split != "auto":
            eng.split_update = bool(int(a.split))
            for p in eng._parts or ():
                p.split_update = eng.split_update
        snap = eng.snapshot()

        def step():
            eng.t = 0
            eng.iterate(0)

        for _ in range(10):
            step()
        eng.join()
        torch.cuda.synchronize()
        eng.restore(snap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            step()
        eng.join()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / a.iters * 1e3
        if a.no_graph:
            print(json.dumps({"scenes": a.scenes, "goals": a.goals, "latency_mode": a.latency, "goal_parts": a.goal_parts, "parts": k, "measure_min": a.measure_min,
                              "ms_per_step_eager": round(eager, 4)}), flush=True)
            del eng
            continue
        # the same steps as one graph of `block` iterations
        eng.restore(snap)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        eng._capturing = True
        try:
            with torch.cuda.graph(g):
                for _ in range(a.block):
                    step()
                eng.join()
        finally:
            eng._capturing = False
        eng.restore(snap)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        reps = max(1, a.iters // a.block)
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / (reps * a.block) * 1e3
        print(json.dumps({"scenes": a.scenes, "goals": a.goals, "latency_mode": a.latency, "goal_parts": a.goal_parts, "parts": k, "ms_per_step_eager": round(eager, 4),
                          "ms_per_step_graph": round(graph, 4)}), flush=True)
        del eng, g


if __name__ == "__main__":
    main()
