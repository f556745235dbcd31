"""How much does the dispatch order of the goal workgroups matter?  (GPU box)  python tools/schedule_experiment.py
Runs the bench workload, reads back the per-goal durations of one launch, builds several schedules on the host and
times the goal-set kernel (HIP events on the dispatch) under each."""
import ctypes as C
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd import _lib  # noqa: E402
from omg_planner_amd.engine import ChompEngine  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 64, 0, False)
    eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD")
    lib = _lib.lib()
    snap = eng.snapshot()

    def run(iters, timed=False):
        eng.restore(snap)
        if timed:
            lib.omgx_timing_enable(1)
        for _ in range(iters):
            eng.t = 0
            eng.iterate(0)
        torch.cuda.synchronize()
        if not timed:
            return None
        buf = (C.c_float * 4096)()
        kinds = (C.c_int32 * 4096)()
        n = lib.omgx_timing_collect(buf, kinds, 4096)
        lib.omgx_timing_enable(0)
        return float(np.mean([buf[i] for i in range(n) if kinds[i] == 0]))

    eng.auto_schedule = False
    eng.schedule = None
    base = run(20, timed=True)
    eng.auto_schedule = True
    run(5)
    auto = run(20, timed=True)
    res0 = {"no schedule, no stamping": round(base * 1e3, 1), "engine.build_schedule (device-built, static)": round(auto * 1e3, 1)}
    eng.schedule = None
    run(1)  # the measuring launch
    eng.auto_schedule = False
    work = eng.work.cpu().numpy().astype(np.int64).reshape(S, G)
    T = work.sum(1)
    out = {"scenes": S, "goals": G, "work_us_mean": float(work.mean() / 100), "work_us_p90": float(np.percentile(work, 90) / 100), "work_us_max": float(work.max() / 100)}
    NS = ((S + 7) // 8) * 8 * G
    items = np.arange(S * G)
    flat = work.reshape(-1)

    def sched_affinity(order_within):
        # scenes -> XCDs by snake over the scene totals; block b (xcd = b & 7) takes the next item of its XCD's list
        order = np.argsort(-T)
        bins = [[] for _ in range(8)]
        for k, q in enumerate(order):
            r = k % 16
            bins[r if r < 8 else 15 - r].append(q)
        lists = []
        for b in bins:
            it = np.concatenate([q * G + np.arange(G) for q in b]) if b else np.zeros(0, np.int64)
            lists.append(order_within(it))
        sched = np.full(NS, -1, np.int64)
        for x, l in enumerate(lists):
            sched[x + 8 * np.arange(len(l))] = l
        return sched

    variants = {
        "natural (no schedule)": None,
        "identity schedule (scene-major through the indirection)": None if S % 8 else (np.arange(S * G).reshape(S // 8, 8, G).transpose(0, 2, 1).reshape(-1)),
        "global LPT, no XCD affinity": items[np.argsort(-flat, kind="stable")],
        "XCD affinity (snake), LPT within XCD": sched_affinity(lambda it: it[np.argsort(-flat[it], kind="stable")]),
        "XCD affinity (snake), scene-major natural": sched_affinity(lambda it: it),
        "XCD affinity (snake), heavy scenes first, goals sorted within scene": sched_affinity(
            lambda it: np.concatenate([q * G + np.argsort(-work[q], kind="stable") for q in sorted(set(it // G), key=lambda q: -T[q])])),
    }
    res = res0
    for name, sc in variants.items():
        if sc is None and name != "natural (no schedule)":
            continue
        if sc is None:
            eng.schedule = None
        else:
            full = np.full(NS, -1, np.int32)
            full[: len(sc)] = sc
            eng.schedule = torch.as_tensor(full, device="cuda:0")
        run(3)
        res[name] = round(run(20, timed=True) * 1e3, 1)
    out["goalset_kernel_us"] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
