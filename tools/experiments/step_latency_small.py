"""Pinned step of 2-4 scenes in latency mode against the layout rule: python tools/experiments/step_latency_small.py"""
import copy, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
import bench
from omg_planner_amd.engine import ChompEngine
dev = torch.device("cuda:0")
for (S, G, n, obj) in [(2, 64, 30, 4), (3, 64, 30, 4), (4, 64, 30, 4), (6, 64, 30, 4), (2, 128, 30, 4), (2, 64, 50, 12), (3, 64, 50, 12)]:
    cfg, model, batch, start, goals = bench.build_workload(S, G, n, 64, 0, False, num_objects=obj)
    res = {}
    for lat in (False, True):
        if lat:
            eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", latency_mode=True)
        else:
            eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=S, device=dev, ol_alg="MD")
        eng.pose_hand_over(True)
        snap = eng.snapshot()
        def step(i):
            if i and i % cfg.optim_steps == 0:
                eng.restore(snap)
            eng.t = 0; eng.iterate(0)
        for i in range(30): step(i)
        best = 1e9
        for _ in range(3):
            eng.join(); torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(200): step(i)
            eng.join(); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
        res["latency" if lat else "rule"] = round(best, 4)
    print(S, G, n, obj, res, flush=True)
