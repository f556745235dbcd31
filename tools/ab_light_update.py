#!/usr/bin/env python3
"""Experiment: what would the pipeline gain if the update launch were LIGHT (workgroups that fit into a goal-set slot)?

The real goal-set launches of k scene ranges on k streams, each followed by stand-ins for the update (tools/light_update_probe.hip:
workgroups that stay resident for a given time with a given LDS footprint): `heavy` = 2 S workgroups x 35 us with 94 KB (today's
k_update_optimize_split, to validate the stand-in against the real 285-290 us), `light` = a per-point stage of 3 S workgroups x
12 us with 12 KB followed by a tail of S workgroups x 25 us with 16 KB.  Trajectories stay fixed (no real update), which does not
change what a goal-set launch costs.
    hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/light_update_probe.hip -o tools/_build/liblight_update_probe.so
    python tools/ab_light_update.py --parts 2|3|4 --mode heavy|light
"""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--mode", default="light", choices=["heavy", "light", "none"])
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    from omg_planner_amd.engine import ChompEngine
    probe = C.CDLL(str(ROOT / "tools" / "_build" / "liblight_update_probe.so"))
    probe.probe_linger.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, 64, 30, 64, 0, False)
    cuts = [a.scenes * k // a.parts for k in range(a.parts + 1)]
    sink = torch.zeros(4096, dtype=torch.float64, device=dev)
    parts = []
    for k in range(a.parts):
        lo, hi = cuts[k], cuts[k + 1]
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            e = ChompEngine(model, batch.subset(lo, hi), cfg, start[lo:hi], goals[lo:hi], device=dev, ol_alg="MD")
            for t in range(4):
                e.t = 0
                e.iterate(0)
        parts.append((e, st))
    torch.cuda.synchronize()

    def step():
        for e, st in parts:
            calls = e._hot[1]
            e.t = 1
            prm = e._learner_params()
            calls.goalset_layer(prm.start_idx, False, e.schedule, None, st.cuda_stream)
            h = C.c_void_p(st.cuda_stream)
            if a.mode == "heavy":
                probe.probe_linger(sink.data_ptr(), 2 * e.S, 35, 94 * 1024, h)
            elif a.mode == "light":
                probe.probe_linger(sink.data_ptr(), 3 * e.S, 12, 12 * 1024, h)
                probe.probe_linger(sink.data_ptr(), e.S, 25, 16 * 1024, h)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"parts": a.parts, "mode": a.mode, "us_per_iteration_of_all_scenes": round(dt / a.iters * 1e6, 1)}))


if __name__ == "__main__":
    main()
