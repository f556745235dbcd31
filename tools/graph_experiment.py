"""Experiment: how much would hipGraph replay of one planner iteration save? (weights frozen at capture)"""
import sys, time, copy
sys.path.insert(0, '.')
import torch, bench
from omg_planner_amd.engine import ChompEngine
cfg, model, batch, start, goals = bench.build_workload(100, 64, 30, 64, 0, False)
eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device="cuda:0", ol_alg="MD")
def step():
    eng.t = 0; eng.iterate(0)
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter() - t0) / 50 * 1e3)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, stream=s):
        step()
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter() - t0) / 50 * 1e3)
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
