"""plan_once.py for any shape: python tools/experiments/plan_once_n.py scenes goals waypoints objects [early]"""
import copy, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
import bench
from omg_planner_amd.engine import ChompEngine
S, G, n, obj = (int(x) for x in sys.argv[1:5])
early = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
cfg, model, batch, start, goals = bench.build_workload(S, G, n, 64, 0, False, num_objects=obj)
import os
if os.environ.get("OMGX_PLAN_LATENCY"):
    eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=torch.device("cuda:0"), ol_alg="MD", latency_mode=True)
    eng.layout_used = {"latency_mode": True}
elif os.environ.get("OMGX_PLAN_GOAL_PARTS"):
    eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=torch.device("cuda:0"), ol_alg="MD", goal_parts=int(os.environ["OMGX_PLAN_GOAL_PARTS"]))
    eng.layout_used = {"goal_parts": int(os.environ["OMGX_PLAN_GOAL_PARTS"])}
else:
    eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=S, device=torch.device("cuda:0"), ol_alg="MD")
if os.environ.get("OMGX_PLAN_PIPELINE"):
    eng.pipeline = int(os.environ["OMGX_PLAN_PIPELINE"])
snap = eng.snapshot()
for rep in range(4):
    eng.restore(snap); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.plan(early_stop=early); torch.cuda.synchronize()
    print("plan ms", (time.perf_counter() - t0) * 1e3, eng.layout_used)
