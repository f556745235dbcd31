"""How long does the HOST take to enqueue a bench step (no device wait), against the step itself?  python tools/experiments/host_rate.py [scenes] [goals] [parts]"""
import copy, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch
import bench
from omg_planner_amd.engine import ChompEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 13
G = int(sys.argv[2]) if len(sys.argv) > 2 else 128
P = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 64, 0, False)
eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=S, device=dev, ol_alg="MD")
if P:
    eng.pipeline = P
eng.pose_hand_over(True)
def step():
    eng.t = 0
    eng.iterate(0)
for _ in range(30):
    step()
eng.join(); torch.cuda.synchronize()
out = {"scenes": S, "goals": G, "pipeline": eng.pipeline}
for n in (20, 50, 200):
    eng.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    eng.join(); torch.cuda.synchronize()
    t2 = time.perf_counter()
    out[f"n{n}"] = {"host_us_per_step": round((t1 - t0) / n * 1e6, 2), "total_us_per_step": round((t2 - t0) / n * 1e6, 2), "drain_us": round((t2 - t1) * 1e6, 1)}
print(json.dumps(out))
