"""Experiment: what would hiding the update launches buy?  The goal-set (+ layer) launches of the two halves alone on two streams,
no update launches at all (trajectories stay put) — the upper bound of any scheme that overlaps the updates with other scenes'
goal-set work — optionally beside N resident stand-ins for update-server workgroups (tools/hold_probe.hip: 512 threads, 94 KB of
LDS each) that take CU slots away from the goal-set kernel.
    python tools/ab_noupdate.py [--hold 32] [--lib omg-planner_amd/csrc/libomg_hip_x.so] [--with-update]"""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hold", type=int, default=0)
    ap.add_argument("--hold-lds", type=int, default=94 * 1024)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--with-update", action="store_true")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--scenes", type=int, default=100)
    a = ap.parse_args()
    if a.lib:
        _lib.LIB_PATH = Path(a.lib).resolve()
    from omg_planner_amd.engine import ChompEngine
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(a.scenes, 64, 30, 64, 0, False)
    cuts = [0, a.scenes // 2, a.scenes]
    engs, streams = [], []
    for k in range(2):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            engs.append(ChompEngine(model, batch.subset(cuts[k], cuts[k + 1]), cfg, start[cuts[k]:cuts[k + 1]], goals[cuts[k]:cuts[k + 1]], device=dev, ol_alg="MD"))
        streams.append(st)
    torch.cuda.synchronize()

    def step():
        for e, st in zip(engs, streams):
            with torch.cuda.stream(st):
                e.t = 0
                if a.with_update:
                    e.iterate(0)
                else:
                    e.update_goal(defer_update=True, with_layer=True)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    hold_stream = torch.cuda.Stream(device=dev)
    if a.hold:
        hl = C.CDLL(str(ROOT / "tools" / "_build" / "libhold.so"))
        hl.hold_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
        est_ms = int(a.iters * 0.3 + 40)
        rc = hl.hold_launch(a.hold, a.hold_lds, est_ms, C.c_void_p(hold_stream.cuda_stream))
        assert rc == 0, rc
        time.sleep(0.005)  # the stand-ins are resident before the timed launches start
    t0 = time.perf_counter()
    for _ in range(a.iters):
        step()
    for st in streams:
        st.synchronize()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(json.dumps({"hold": a.hold, "hold_lds": a.hold_lds, "with_update": a.with_update, "lib": Path(_lib.LIB_PATH).name,
                      "us_per_step": round(dt / a.iters * 1e6, 1)}))


if __name__ == "__main__":
    main()
