"""Experiment (round 4): what would overlapping a half's update launch with the SAME half's goal-set launch buy, at best?
Timing only — the data dependency is ignored (the update launch of iteration k runs on a second stream beside the goal-set launch
of iteration k instead of behind it; the goal-set launch of iteration k + 1 waits for both), so the results are garbage and the
time is a LOWER bound for any scheme that starts the update workgroups before the goal-set launch has ended (tickets per scene,
a gate at 80 % of the launch, ...).   python tools/ab_overlap_update.py [--mode serial|overlap] [--delay-frac 0.0]"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="overlap", choices=["serial", "overlap"])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--goals", type=int, default=64)
    a = ap.parse_args()
    from omg_planner_amd.engine import ChompEngine
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, 30, 64, 0, False)
    cuts = [0, a.scenes // 2, a.scenes]
    engs, sa, su = [], [], []
    for k in range(2):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            engs.append(ChompEngine(model, batch.subset(cuts[k], cuts[k + 1]), cfg, start[cuts[k]:cuts[k + 1]], goals[cuts[k]:cuts[k + 1]], device=dev, ol_alg="MD"))
        sa.append(st)
        su.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()

    def step():
        for e, A, U in zip(engs, sa, su):
            e.t = 0
            if a.mode == "serial":
                with torch.cuda.stream(A):
                    e.iterate(0)
                continue
            with torch.cuda.stream(A):
                prm = e.update_goal(defer_update=True, with_layer=True)   # the goal-set + layer launch
            with torch.cuda.stream(U):
                e._schedule()
                e._step(True, prm)                                        # the update launch, beside it (no dependency: timing only)
            A.wait_stream(U)                                              # the next goal-set launch waits for both
            U.wait_stream(A)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"mode": a.mode, "scenes": a.scenes, "goals": a.goals, "us_per_step": round(dt / a.iters * 1e6, 1)}))


if __name__ == "__main__":
    main()
