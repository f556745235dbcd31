"""One scene's plan (latency mode) as one HIP graph for a few tilings of the goal-set / layer launch: goal parts x layer block.
    python tools/ab_lat_parts.py"""
import copy
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd.engine import ChompEngine  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(1, 64, 30, 64, 0, False)
    for gp, lb, lg in ((4, 4, 10), (8, 4, 10), (2, 4, 10), (4, 2, 10), (4, 8, 10), (8, 2, 10), (4, 4, 5)):
        ChompEngine.LAT_GOAL_PARTS, ChompEngine.LAT_LAYER_BLOCK, ChompEngine.LAT_LAYER_LINK_GROUPS = gp, lb, lg
        e = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", latency_mode=True)
        fresh = e.snapshot()
        pg = e.capture_plan(early_stop=True)
        best = []
        for _ in range(5):
            e.restore(fresh)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pg.replay()
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t0) * 1e3)
        print(json.dumps({"goal_parts": gp, "layer_block": lb, "layer_link_groups": lg, "ms_per_plan_graph_min/median": [round(min(best), 4), round(sorted(best)[2], 4)]}), flush=True)


if __name__ == "__main__":
    main()
