"""Experiment (round 4): the ORDER of a mid-size launch's goal workgroups inside an XCD.
The measured schedule lists an XCD's items scene by scene (goals longest first inside a scene).  When the launch is only a round or
two of the chip's workgroup slots, its span is set by what starts LAST; here the same items per XCD are re-ordered longest first
across the XCD's scenes (host-side, from the measured durations) and the step is timed with both.
    python tools/ab_schedule_order.py --scenes 13 --goals 128 [--parts 1]"""
import argparse
import copy
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=13)
    ap.add_argument("--goals", type=int, default=128)
    ap.add_argument("--parts", type=int, default=1)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    from omg_planner_amd.engine import ChompEngine
    ChompEngine.MEASURE_MIN_ITEMS = 64
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, 30, 64, 0, False)
    cuts = [a.scenes * k // a.parts for k in range(a.parts + 1)]
    engs, sa = [], []
    for k in range(a.parts):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            engs.append(ChompEngine(model, batch.subset(cuts[k], cuts[k + 1]), copy.deepcopy(cfg), start[cuts[k]:cuts[k + 1]], goals[cuts[k]:cuts[k + 1]], device=dev, ol_alg="MD"))
        sa.append(st)

    def step():
        for e, A in zip(engs, sa):
            e.t = 0
            with torch.cuda.stream(A):
                e.iterate(0)

    def timed():
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        out = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(a.iters):
                step()
            torch.cuda.synchronize()
            out.append(round((time.perf_counter() - t0) / a.iters * 1e3, 4))
        return out

    snaps = [e.snapshot() for e in engs]
    res = {"scenes": a.scenes, "goals": a.goals, "parts": a.parts}
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    assert all(e._measured for e in engs), "no measured schedule"
    for e, s in zip(engs, snaps):
        e.restore(s)
    res["scene_major"] = timed()
    for e, s in zip(engs, snaps):
        e.restore(s)
        sched = e.schedule.cpu().numpy().reshape(-1, 8).copy()
        work = e.work[: e.S * e.G].cpu().numpy().astype(np.int64)
        new = np.full_like(sched, -1)
        for x in range(8):
            items = sched[:, x][sched[:, x] >= 0]
            order = np.lexsort((items, -work[items]))  # longest first, ties by index
            new[: len(items), x] = items[order]
        e.schedule = torch.as_tensor(new.reshape(-1), device=dev)
        e._hot = None
    torch.cuda.synchronize()
    res["longest_first_in_xcd"] = timed()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
