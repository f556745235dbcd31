"""One scene in latency mode: plan ms against the layer's tiling (class attributes LAT_LAYER_LINK_GROUPS / LAT_LAYER_BLOCK / LAT_GOAL_PARTS)."""
import copy, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
import bench
from omg_planner_amd.engine import ChompEngine
dev = torch.device("cuda:0")
n, obj = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (30, 4)
cfg, model, batch, start, goals = bench.build_workload(1, 64, n, 64, 0, False, num_objects=obj)
base = (ChompEngine.LAT_GOAL_PARTS, ChompEngine.LAT_LAYER_LINK_GROUPS, ChompEngine.LAT_LAYER_BLOCK)
for gp, lg, cb in [base, (4, 10, 8), (4, 10, 2), (4, 10, 15), (4, 5, 4), (4, 10, 6), (8, 10, 4), (2, 10, 4)]:
    ChompEngine.LAT_GOAL_PARTS, ChompEngine.LAT_LAYER_LINK_GROUPS, ChompEngine.LAT_LAYER_BLOCK = gp, lg, cb
    eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", latency_mode=True)
    snap = eng.snapshot()
    best, bestg = 1e9, 1e9
    for rep in range(6):
        eng.restore(snap); torch.cuda.synchronize(); t0 = time.perf_counter(); eng.plan(early_stop=False); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    eng.restore(snap)
    fresh = eng.snapshot()
    g = eng.capture_plan(early_stop=False)
    for rep in range(6):
        eng.restore(fresh); torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        bestg = min(bestg, (time.perf_counter() - t0) * 1e3)
    print(f"n={n} objects={obj + 1} goal_parts {gp} layer {lg} x blocks of {cb}: plan {best:.3f} ms, as one graph {bestg:.3f}", flush=True)
