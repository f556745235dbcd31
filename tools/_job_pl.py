import sys, time, copy
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch, numpy as np
import bench
from omg_planner_amd.engine import ChompEngine
S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg, model, batch, start, goals = bench.build_workload(S, 64, 30, 64, 0, False)
res = {}
engs = {}
for hot in (True, False):
    e = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, device="cuda:0", ol_alg="MD")
    engs[hot] = (e, e.snapshot())
for rep in range(6):
    for hot in (True, False):
        ChompEngine.HOT_FIXED_GOAL = hot
        e, fresh = engs[hot]
        for early in (False, True):
            e.restore(fresh); torch.cuda.synchronize(); t0 = time.perf_counter(); e.plan(early_stop=early); torch.cuda.synchronize()
            res.setdefault((hot, early), []).append((time.perf_counter() - t0) * 1e3)
for k, v in res.items():
    print("hot_fixed=%s early_stop=%s: best %.3f median %.3f ms" % (k[0], k[1], min(v[1:]), float(np.median(v[1:]))))
a = engs[True][0]; b = engs[False][0]
print("same bits:", all(np.array_equal(getattr(a, k).cpu().numpy(), getattr(b, k).cpu().numpy(), equal_nan=True) for k in ("traj", "info", "goal_idx", "cost_traj", "grad")))
