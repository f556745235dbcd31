"""Which entries differ (if any) between K launched iterations and the persistent launch: per tensor the count, the largest difference
and — for the learner's state [sum_costs | p | experts_p x5 | q | expert costs] — the block."""
import copy
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import os  # noqa: E402

from omg_planner_amd import _lib  # noqa: E402

if os.environ.get("OMGX_LIB"):
    _lib.LIB_PATH = ROOT / "omg-planner_amd" / "csrc" / os.environ["OMGX_LIB"]
import bench  # noqa: E402
from omg_planner_amd.engine import ChompEngine  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    out = {}
    for alg, S, G, K in (("MD", 6, 16, 1), ("MD", 6, 16, 12), ("FTL", 3, 9, 12), ("Exp", 4, 9, 6)):
        cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 24, 3, False)
        mk = lambda: ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=alg)
        a, b = mk(), mk()
        for e in (a, b):
            e.select_initial_goal()
            e.pose_hand_over(True)
        for t in range(K):
            a.iterate(t)
        b.run_persistent(range(K))
        torch.cuda.synchronize()
        rec = {}
        for k in ("traj", "info", "learner_state", "goal_idx", "grad", "cost_traj", "cost_vec", "goal_cost", "pot", "pgrad", "end_pose", "wp_pose"):
            x, y = getattr(a, k).cpu().numpy().astype(np.float64), getattr(b, k).cpu().numpy().astype(np.float64)
            d = np.abs(x - y)
            if (x != y).any():
                rec[k] = {"count": int((x != y).sum()), "max": float(d.max()), "rel": float((d / np.maximum(np.abs(x), 1e-300)).max())}
                if k == "learner_state":
                    idx = np.argwhere(x != y)[:, 1]
                    blocks = {"sum_costs": (0, G), "p": (G, 2 * G), "experts_p": (2 * G, 7 * G), "q": (7 * G, 7 * G + 5), "ecost": (7 * G + 5, 7 * G + 10)}
                    rec[k]["blocks"] = {n: int(((idx >= lo) & (idx < hi)).sum()) for n, (lo, hi) in blocks.items()}
                if k == "info":
                    rec[k]["columns"] = sorted(set(int(c) for c in np.argwhere(x != y)[:, 1]))
        out[f"{alg}_{S}x{G}_K{K}"] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
