"""Experiment: a whole ChompEngine.plan captured as ONE HIP graph (torch.cuda.CUDAGraph) against the eager plan.
    python tools/graph_experiment.py [scenes] [goals] [early_stop 0|1]
Replaying the graph re-runs the plan from whatever state the engine's tensors hold (kernel arguments are frozen at capture)."""
import copy
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

import bench
from omg_planner_amd.engine import ChompEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
early = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 64, 0, False)


def fresh():
    return ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device="cuda:0", ol_alg="MD")


best = float("inf")
for _ in range(3):
    e = fresh()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.plan(early_stop=early)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) * 1e3)
ref = e.info.cpu().numpy().copy()
print(f"eager plan {best:.2f} ms")

eng = fresh()
snap0 = eng.snapshot()
eng.plan(early_stop=early)          # warm-up: workspaces, schedules, parts, side streams
eng.restore(snap0)
torch.cuda.synchronize()
if early and S <= 4:
    eng._plan_all_done = lambda *a: False  # the host-side look at the mask cannot be captured
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        eng.plan(early_stop=early)
        eng.join()
    torch.cuda.synchronize()
    times = []
    for _ in range(4):
        eng.restore(snap0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    import numpy as np
    same = np.array_equal(eng.info.cpu().numpy(), ref, equal_nan=True)
    print(f"graph replay {min(times):.2f} ms (all: {[round(x, 2) for x in times]}), results equal to the eager plan: {same}")
except Exception as ex:  # noqa: BLE001
    print("graph capture failed:", repr(ex)[:400])
