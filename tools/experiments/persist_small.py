"""Repro aid (round 6, fuzz campaign r06final): tiny plans through the persistent launch with dedicated update CUs against the launched iterations;
prints the tensors that differ.  python tools/experiments/persist_small.py"""
import copy, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import torch
import bench
from omg_planner_amd.engine import ChompEngine

CMP = ("traj", "info", "learner_state", "goal_idx", "grad", "cost_traj", "pot", "pgrad", "col", "goal_cost", "goal_col", "end", "goal_rows", "goal_point")
dev = torch.device("cuda:0")
bad = 0
for (S, G, n, alg, proj) in [(1, 1, 12, "FTL", False), (1, 5, 20, "Exp", False), (1, 2, 41, "FTL", True), (2, 2, 12, "MD", True), (1, 8, 30, "MD", True), (3, 1, 12, "FTL", False)]:
    for ucu in (0, 1, 2):
        for per_launch in (1, 3):
            cfg, model, batch, start, goals = bench.build_workload(S, G, n, 20, 3, False, num_objects=3)
            cfg.goal_set_proj = proj
            mk = lambda: ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=alg)
            a, b = mk(), mk()
            for e in (a, b):
                e.select_initial_goal(); e.pose_hand_over(True)
            msg = []
            for t0 in range(0, 6, per_launch):
                ts = list(range(t0, t0 + per_launch))
                for t in ts:
                    a.iterate(t)
                b.run_persistent(ts, update_cus=ucu)
                torch.cuda.synchronize()
                for k in CMP:
                    x, y = getattr(a, k, None), getattr(b, k, None)
                    if x is None:
                        continue
                    if not torch.equal(x, y):
                        d = (x.double() - y.double()).abs()
                        msg.append(f"t={ts} {k}: max {float(d.max()):.3e} at {int(d.argmax())} of {tuple(x.shape)}")
                if msg:
                    break
            st = b.persistent_status()
            print(f"S={S} G={G} n={n} {alg} proj={proj} update_cus={ucu} per_launch={per_launch}: {'ok' if not msg else 'DIFF'} status {st}")
            for m in msg[:8]:
                print("    ", m)
            bad += bool(msg)
print("bad", bad)
