"""Experiment (round 4): what would an update workgroup of another SHAPE do to the step?  Timing only.
The two halves of the bench batch iterate on two streams like the pipeline does; the update launch is replaced by 2 S/2 workgroups of
tools/spin_update.hip that hold (threads, LDS bytes, VGPRs) for `usec` and compute nothing (the trajectories therefore stay where they
are: compare the stand-in shapes with each other and with `none`, not with `real`).
    python tools/ab_light_update.py --shape none | real | THREADSxLDSxVGPRSxUSEC  (e.g. 512x96256x151x27, 256x31744x151x40)"""
import argparse
import ctypes
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
    ap.add_argument("--shape", default="none")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--offset-usec", type=int, default=0, help="part k's stream starts every timed region k x this late (one idle workgroup)")
    a = ap.parse_args()
    from omg_planner_amd.engine import ChompEngine
    dev = torch.device("cuda:0")
    spin = ctypes.CDLL(str(ROOT / "tools" / "experiments" / "spin_update.so"))
    spin.spin_launch.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p]
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, 30, 64, 0, False)
    cuts = [a.scenes * k // a.parts for k in range(a.parts + 1)]
    engs, sa = [], []
    for k in range(a.parts):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            engs.append(ChompEngine(model, batch.subset(cuts[k], cuts[k + 1]), cfg, start[cuts[k]:cuts[k + 1]], goals[cuts[k]:cuts[k + 1]], device=dev, ol_alg="MD"))
        sa.append(st)
    torch.cuda.synchronize()
    shape = None if a.shape in ("none", "real") else [int(x) for x in a.shape.split("x")]

    def step():
        for e, A in zip(engs, sa):
            e.t = 0
            with torch.cuda.stream(A):
                if a.shape == "real":
                    e.iterate(0)
                    continue
                e.update_goal(defer_update=True, with_layer=True)   # the goal-set + layer launch
                if shape:
                    rc = spin.spin_launch(2 * e.S, shape[0], shape[1], shape[2], shape[3], A.cuda_stream)
                    assert rc == 0, rc

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    out = []
    for _ in range(3):
        t0 = time.perf_counter()
        if a.offset_usec:
            for k in range(1, a.parts):
                spin.spin_launch(1, 64, 0, 79, k * a.offset_usec, sa[k].cuda_stream)
        for _ in range(a.iters):
            step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / a.iters * 1e3)
    print(json.dumps({"shape": a.shape, "scenes": a.scenes, "goals": a.goals, "parts": a.parts, "offset_usec": a.offset_usec, "ms_per_step": [round(x, 4) for x in out]}), flush=True)


if __name__ == "__main__":
    main()
