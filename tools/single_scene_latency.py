import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np, torch, bench, copy
from omg_planner_amd.engine import ChompEngine
cfg, model, batch, start, goals = bench.build_workload(1, 64, 30, 64, 0, False)
for rep in range(3):
    eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device="cuda:0", ol_alg="MD")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.plan(early_stop=False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("single-scene plan ms", (t1 - t0) * 1e3, "host-only? iterations 71")
eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device="cuda:0", ol_alg="MD")
t0 = time.perf_counter()
for t in range(70): eng.iterate(t)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host ms", (t1-t0)*1e3, "total ms", (t2-t0)*1e3)
