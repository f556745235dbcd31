import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import torch, bench
from omg_planner_amd.engine import ChompEngine
# tiny workload by default: the GPU finishes long before the host has issued the next iteration -> pure host cost
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
alg = sys.argv[3] if len(sys.argv) > 3 else "MD"
cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 64, 0, False)
eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg=alg)
for _ in range(5):
    eng.t = 0; eng.iterate(0)
torch.cuda.synchronize()
# pure host time: launch 50 iterations without syncing, time the host loop
t0 = time.perf_counter()
for _ in range(50):
    eng.t = 0; eng.iterate(0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host ms/iter", (t1 - t0) / 50 * 1e3, "total ms/iter", (t2 - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    eng.t = 0; eng.iterate(0)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
