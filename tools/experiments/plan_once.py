"""A few plans of the bench batch (eager), for a kernel trace: python tools/experiments/plan_once.py [scenes] [goals] [early_stop]"""
import copy, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
import bench
from omg_planner_amd.engine import ChompEngine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
early = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 64, 0, False)
eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=S, device=torch.device("cuda:0"), ol_alg="MD")
snap = eng.snapshot()
import time
for rep in range(6):
    eng.restore(snap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.plan(early_stop=early)
    torch.cuda.synchronize()
    print("plan ms", (time.perf_counter() - t0) * 1e3)
