import sys, json, time, cProfile, pstats, copy
sys.path.insert(0, '.')
import torch, bench
import numpy as np
dev = torch.device("cuda:0")
print(json.dumps(bench.drop_in_plan_timing(dev, "MD"))[:60])
# profile only warm plans: monkeypatch perf_counter brackets by profiling the whole thing but sorting by cumulative within loop functions
pr = cProfile.Profile()
import omg_planner_amd.optimizer as O, omg_planner_amd.online_learner as L
orig_opt, orig_upd = O.Optimizer.optimize, L.Learner.update_goal
def opt(self, *a, **k):
    pr.enable()
    try: return orig_opt(self, *a, **k)
    finally: pr.disable()
def upd(self, *a, **k):
    pr.enable()
    try: return orig_upd(self, *a, **k)
    finally: pr.disable()
O.Optimizer.optimize, L.Learner.update_goal = opt, upd
bench.drop_in_plan_timing(dev, "MD", reps=3)
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(32)
O.Optimizer.optimize, L.Learner.update_goal = orig_opt, orig_upd
# engine init profile
cfg, model, batch, start, goals = bench.build_workload(100, 64, 30, 64, 0, False, device=dev)
from omg_planner_amd.engine import ChompEngine
e = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
torch.cuda.synchronize()
pr2 = cProfile.Profile(); pr2.enable()
t0=time.perf_counter()
e = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
torch.cuda.synchronize()
print("engine init ms", (time.perf_counter()-t0)*1e3)
pr2.disable()
pstats.Stats(pr2).sort_stats("tottime").print_stats(14)
