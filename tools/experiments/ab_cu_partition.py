#!/usr/bin/env python3
"""Experiment (GPU box): the pipeline's goal-set launches on one partition of the CUs and its update launches on the rest
(hipExtStreamCreateWithCUMask; mask bit i = CU i / 8 of XCD i % 8, tools/experiments/cu_mask_probe.hip), so that an update
workgroup (94 KB of LDS, eight waves of 151 VGPRs) never waits for a CU the other parts' goal-set workgroups keep refilling.
Every part gets a goal-set stream and an update stream, ordered by two events per iteration (raw HIP calls: a torch event pair
costs 10 us of host time per use).

    python tools/experiments/ab_cu_partition.py [--reserve R] [--parts K] [--update-anywhere] [--steps N]
R = CUs per XCD kept free of goal-set workgroups (0: plain streams, the shipped pipeline); --update-anywhere: the update streams
carry no mask (they may also take CUs of the goal-set partition).  Prints ms per step over N steps (median of 5 regions).
"""
import argparse
import ctypes as C
import json
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reserve", type=int, default=2)
    ap.add_argument("--parts", type=int, default=3)
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--update-anywhere", action="store_true")
    ap.add_argument("--side-only", action="store_true", help="update on a second stream per part, no masks at all")
    a = ap.parse_args()
    from omg_planner_amd.engine import ChompEngine
    dev = torch.device("cuda:0")
    torch.cuda.init()
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]

    def masked_stream(words):
        st = C.c_void_p()
        arr = (C.c_uint32 * 8)(*words)
        assert hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, arr) == 0
        return st.value

    def plain_stream():
        st = C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0  # hipStreamNonBlocking
        return st.value

    def event():
        ev = C.c_void_p()
        assert hip.hipEventCreateWithFlags(C.byref(ev), 2) == 0  # hipEventDisableTiming
        return ev.value

    def rec(ev, stream):
        assert hip.hipEventRecord(ev, stream) == 0

    def wait(stream, ev):
        assert hip.hipStreamWaitEvent(stream, ev, 0) == 0

    nbits = 8 * a.reserve
    low = [0] * 8
    for b in range(nbits):
        low[b // 32] |= 1 << (b % 32)
    high = [(~w) & 0xffffffff for w in low]

    cfg, model, batch, start, goals = bench.build_workload(a.scenes, a.goals, 30, 64, 0, False)
    use_masks = a.reserve > 0 and not a.side_only
    gs_streams = [torch.cuda.ExternalStream(masked_stream(high) if use_masks else plain_stream(), device=dev) for _ in range(a.parts)]
    with torch.cuda.stream(gs_streams[0]):
        eng = ChompEngine(model, batch, cfg, start, goals, device=dev, ol_alg="MD")
        eng.pipeline = a.parts
        parts = eng._get_parts(a.parts)
        for i, part in enumerate(parts[1:], 1):
            part.stream = gs_streams[i]
        if a.reserve > 0 or a.side_only:
            from omg_planner_amd import ops
            side = {}  # a part's stream handle -> (its update stream, two events)
            for i, part in enumerate(parts):
                up = plain_stream() if (a.update_anywhere or a.side_only) else masked_stream(low)
                side[gs_streams[i].cuda_stream] = (up, event(), event())
            plain_update = ops.IterationCalls.update

            def update(self, lp, prm, split, ticket, stop, stream):  # the update launch on the part's second stream, ordered by events
                up, ev_a, ev_b = side[stream]
                rec(ev_a, stream); wait(up, ev_a)
                plain_update(self, lp, prm, split, ticket, stop, up)
                rec(ev_b, up); wait(stream, ev_b)

            ops.IterationCalls.update = update

        snap = eng.snapshot()
        count = [0]

        def step():  # bench.py's step: every 50 steps back to the fresh plan, the goal-set window pinned at n waypoints
            if count[0] and count[0] % cfg.optim_steps == 0:
                eng.restore(snap)
            count[0] += 1
            eng.t = 0
            eng.iterate(0)

        def region(n):
            eng.join()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            eng.join()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        for _ in range(12):  # through the measuring launch and the measured schedule of every part
            step()
        region(20)
        ms = [region(a.steps) for _ in range(5)]
    print(json.dumps({"reserve_per_xcd": a.reserve, "parts": a.parts, "update_anywhere": a.update_anywhere, "side_only": a.side_only,
                      "ms_per_step_median": statistics.median(ms), "ms_per_step_all": [round(x, 4) for x in ms]}))


if __name__ == "__main__":
    main()
