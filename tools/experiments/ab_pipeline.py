#!/usr/bin/env python3
"""Experiment: software pipeline of two half-batches on two streams against one engine for the whole batch (needs a GPU).

The step of one engine is goal-set launch (fills the GPU, ~275 us) -> update launch (200 workgroups, latency-bound, ~35 us): the
tail of the first, the second and the ramp-up of the next launch leave the GPU underused for ~80 us per iteration.  Two
independent engines of half the scenes, enqueued alternately on two streams (optionally with different priorities), could
overlap one half's underused phases with the other half's goal-set launch.

    python tools/ab_pipeline.py --mode single|dual [--prio] [--scenes 100] [--iters 200]
Prints one JSON line: scene-iterations per second over the timed block.
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
    ap.add_argument("--mode", default="single", choices=["single", "dual", "quad"])
    ap.add_argument("--prio", action="store_true", help="dual: first stream high priority, second low")
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--offset", action="store_true", help="dual: start the second engine half an iteration late")
    a = ap.parse_args()
    from omg_planner_amd.engine import ChompEngine
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, 30, 64, 0, False)
    parts = {"single": 1, "dual": 2, "quad": 4}[a.mode]
    cuts = [a.scenes * k // parts for k in range(parts + 1)]
    engs, streams = [], []
    for k in range(parts):
        lo, hi = cuts[k], cuts[k + 1]
        prio = (-1 if k == 0 else 0) if a.prio else 0
        st = torch.cuda.Stream(device=dev, priority=prio) if parts > 1 else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            engs.append(ChompEngine(model, batch.subset(lo, hi), cfg, start[lo:hi], goals[lo:hi], device=dev, ol_alg="MD"))
        streams.append(st)
    torch.cuda.synchronize()

    def step():
        for e, st in zip(engs, streams):
            with torch.cuda.stream(st):
                e.t = 0
                e.iterate(0)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"mode": a.mode, "prio": a.prio, "scenes": a.scenes, "us_per_iteration_of_all_scenes": round(dt / a.iters * 1e6, 1),
                      "scene_iterations_per_s": round(a.scenes * a.iters / dt)}))


if __name__ == "__main__":
    main()
