"""Which of two goal-set launches that run at once gets the chip?  (round 4: why the pipeline's halves run in step)
Two engines (50 and 51 scenes) iterate on two streams like the pipeline's halves; the instrumented library
    make -C omg-planner_amd/csrc BUILD=build_clk OUT=libomg_hip_clk.so EXTRA=-DOMGX_GS_CLOCK=1
stamps every goal workgroup's start / end (100 MHz); printed: per 10 us slice of the last iterations, the resident goal workgroups of
each half and how many each half STARTED in the slice.
    python tools/gs_two_queue_clock.py [--shape real|none|THREADSxLDSxVGPRSxUSEC] [--offset-usec 90] [--iters 6]"""
import argparse
import ctypes as C
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd import _lib  # noqa: E402

_lib.LIB_PATH = ROOT / "omg-planner_amd" / "csrc" / os.environ.get("OMGX_CLK_LIB", "libomg_hip_clk.so")
from omg_planner_amd.engine import ChompEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="real")
    ap.add_argument("--offset-usec", type=int, default=0)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--slice-us", type=float, default=10.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    spin = C.CDLL(str(ROOT / "tools" / "spin_update.so"))
    spin.spin_launch.argtypes = [C.c_int] * 5 + [C.c_void_p]
    cfg, model, batch, start, goals = bench.build_workload(101, 64, 30, 64, 0, False)
    cuts = [0, 50, 101]
    engs, sa = [], []
    for k in range(2):
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
                e.update_goal(defer_update=True, with_layer=True)
                if shape:
                    assert spin.spin_launch(2 * e.S, shape[0], shape[1], shape[2], shape[3], A.cuda_stream) == 0

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    if a.offset_usec:
        spin.spin_launch(1, 64, 0, 79, a.offset_usec, sa[1].cuda_stream)
    for _ in range(a.iters):
        step()
    torch.cuda.synchronize()
    lib = _lib.lib()
    n = 1 << 15
    buf = (C.c_ulonglong * (8 * n))()
    lib.omgx_debug_gs_wg.argtypes = [C.c_void_p, C.c_int]
    assert lib.omgx_debug_gs_wg(buf, n) == 0
    w = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
    half = (np.arange(n) >> 14) & 1
    nlayer = 7 * 5 * 8
    idx = np.arange(n) & 0x3fff
    ok = (w[:, 4] > w[:, 0]) & (w[:, 0] > 0) & (idx >= nlayer)
    tmax = w[ok, 4].max()
    ok &= w[:, 0] > tmax - 40000  # the last 400 us: the last launch of each half (stamps of earlier launches are overwritten by later ones)
    t0 = w[ok, 0].min()
    st, en = (w[:, 0] - t0) / 100.0, (w[:, 4] - t0) / 100.0
    out = {"shape": a.shape, "offset_usec": a.offset_usec}
    for h in (0, 1):
        m = ok & (half == h)
        out[f"half{h}"] = {"wgs": int(m.sum()), "first_start": float(st[m].min()), "last_start": float(st[m].max()), "last_end": float(en[m].max()),
                           "life_mean": float((en - st)[m].mean())}
    print(json.dumps(out))
    span = en[ok].max()
    print("  t_us   resident h0   h1 | started h0   h1")
    for b in range(int(span / a.slice_us) + 1):
        lo, hi = b * a.slice_us, (b + 1) * a.slice_us
        row = []
        for h in (0, 1):
            m = ok & (half == h)
            row.append((np.minimum(en[m], hi).clip(lo) - np.maximum(st[m], lo).clip(None, hi)).sum() / a.slice_us)
        s0 = int((ok & (half == 0) & (st >= lo) & (st < hi)).sum())
        s1 = int((ok & (half == 1) & (st >= lo) & (st < hi)).sum())
        print(f"{lo:6.0f}   {row[0]:11.0f} {row[1]:5.0f} | {s0:9d} {s1:5d}")


if __name__ == "__main__":
    main()
