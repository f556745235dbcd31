#!/usr/bin/env python3
"""Does re-measuring the goal workgroups' durations under the schedule they produced (and rebuilding it) balance the XCDs better
than the single measurement on the uniform schedule?  Times `iters` bench steps after 0, 1, 2 extra measure-and-rebuild rounds.

    python tools/ab_reschedule.py [scenes] [pipeline parts]
"""
import copy
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

import bench
from omg_planner_amd.engine import ChompEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg, model, batch, start, goals = bench.build_workload(S, 64, 30, 64, 0, False)


def run(extra_rounds, iters=200):
    eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device="cuda:0", ol_alg="MD")
    eng.pipeline = parts
    snap = eng.snapshot()

    def step(k):
        if k and k % 50 == 0:
            eng.restore(snap)
        eng.t = 0
        eng.iterate(0)

    for k in range(6):
        step(k)
    for r in range(extra_rounds):  # measure under the current schedule, rebuild, settle
        for e in (eng._parts or [eng]):
            e._measured = False
            e._gs_launches = 1
        for k in range(4):
            step(k + 1)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(iters):
            step(k + 1)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    return best


for extra in (0, 1, 2, 0, 1, 2):
    print(f"extra measure-and-rebuild rounds {extra}: {run(extra):.4f} ms per step", flush=True)
