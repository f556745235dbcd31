"""Per-workgroup timeline of the goal-set kernel: where a goal workgroup's time goes (the three prologue stages, first /
last wave leaving the main loop), how the launch fills the chip over time, when each XCD finishes.

Needs the instrumented variant of the library (never the shipped one):
    make -C omg-planner_amd/csrc BUILD=build_clk OUT=libomg_hip_clk.so EXTRA=-DOMGX_GS_CLOCK=1
Run on the GPU box:  python tools/gs_phase_clock.py [scenes] [goals]
Every workgroup stamps the 100 MHz realtime clock at its phase boundaries (one scalar clock read each: no measurable effect on
the kernel, unlike in-loop shader-clock reads, which cost ~1 us apiece and slowed it 12x when tried).
"""
import ctypes as C
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd import _lib  # noqa: E402

import os
_lib.LIB_PATH = ROOT / "omg-planner_amd" / "csrc" / os.environ.get("OMGX_CLK_LIB", "libomg_hip_clk.so")  # experiment variants of the instrumented build
from omg_planner_amd.engine import ChompEngine  # noqa: E402

def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    NW = int(sys.argv[3]) if len(sys.argv) > 3 else 30  # waypoints (the goal-set window at t = 0)
    cfg, model, batch, start, goals = bench.build_workload(S, G, NW, 64, 0, False)
    eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD")
    lib = _lib.lib()
    iters = 10
    for _ in range(3 + iters):
        eng.t = 0
        eng.iterate(0)
    torch.cuda.synchronize()
    out = {"scenes": S, "goals": G, "launches": iters}
    # per-workgroup timeline of the LAST launch: 100 MHz realtime stamps at entry / after the prologue / at exit + hardware id
    import numpy as np
    nwg = ((S + 7) // 8) * 5 * 8 + max(((S + 7) // 8) * G * 8, 0 if eng.schedule is None else int(eng.schedule.numel()))
    base = (S & 1) << 14  # the instrumented kernel keeps the stamps of a launch with an odd number of scenes apart (tools/experiments/gs_two_queue_clock.py)
    wg = (C.c_ulonglong * (8 * (base + nwg)))()
    lib.omgx_debug_gs_wg.argtypes = [C.c_void_p, C.c_int]
    assert lib.omgx_debug_gs_wg(wg, base + nwg) == 0
    w = np.frombuffer(wg, dtype=np.uint64).reshape(base + nwg, 8)[base:].copy()
    # stamps of the LAST launch only: blocks that were empty in it may still carry an earlier launch's stamps (other schedule)
    ran = (w[:, 4] > 0) & (w[:, 0] + np.uint64(100000) > w[:, 4].max()) & (w[:, 4] > w[:, 0])
    t0 = w[ran, 0].min()
    st, t_sc, t_ch, pro, en = [(w[:, k].astype(np.int64) - int(t0)) / 100.0 for k in range(5)]  # microseconds
    nlayer = ((S + 7) // 8) * 5 * 8
    is_layer = np.arange(nwg) < nlayer
    xcc = (w[:, 7] >> np.uint64(32)).astype(np.int64) & 0xF
    hw = w[:, 7].astype(np.int64) & 0xFFFFFFFF
    cu = (hw >> 8) & 0xF
    se = (hw >> 13) & 0x7
    dur = en - st
    out["timeline_us"] = {
        "kernel_span": float(en[ran].max()),
        "last_start": float(st[ran].max()),
        "goal_wg_duration_mean/p50/p90/max": [float(x) for x in (dur[ran & ~is_layer].mean(), np.percentile(dur[ran & ~is_layer], 50), np.percentile(dur[ran & ~is_layer], 90), dur[ran & ~is_layer].max())],
        "goal_wg_prologue_mean": float((pro - st)[ran & ~is_layer].mean()),
        "goal_wg_sincos/chain/rowcull_mean": [float(x[ran & ~is_layer].mean()) for x in (t_sc - st, t_ch - t_sc, pro - t_ch)],
        "goal_wg_first/last_wave_leaves_main_loop_after_prologue_mean": [float(((w[:, 5].astype(np.int64) - int(t0)) / 100.0 - pro)[ran & ~is_layer].mean()),
                                                                           float(((w[:, 6].astype(np.int64) - int(t0)) / 100.0 - pro)[ran & ~is_layer].mean())],
        "layer_wg_duration_mean/max": [float(dur[ran & is_layer].mean()), float(dur[ran & is_layer].max())],
        "finish_per_xcc": {int(x): float(en[ran & (xcc == x)].max()) for x in np.unique(xcc[ran])},
        "wgs_per_xcc": {int(x): int((ran & (xcc == x)).sum()) for x in np.unique(xcc[ran])},
        "blockidx_mod8_equals_xcc": float((xcc[ran] == (np.arange(nwg)[ran] & 7)).mean()),
    }
    # the clock the CUs really ran at during the launch: shader-clock cycles per 100 MHz tick over each goal workgroup's lifetime
    fr = (C.c_ulonglong * (2 * nwg))()
    lib.omgx_debug_gs_freq.argtypes = [C.c_void_p, C.c_int]
    if lib.omgx_debug_gs_freq(fr, nwg) == 0:
        f = np.array(list(fr), dtype=np.float64).reshape(nwg, 2)
        ok = ran & ~is_layer & (f[:, 1] > 1000)
        ghz = f[ok, 0] / f[ok, 1] * 0.1
        out["shader_clock_ghz_p10/p50/p90"] = [round(float(np.percentile(ghz, q)), 3) for q in (10, 50, 90)]
    # resident workgroups over time (all CUs): fraction of the 1280 slots (5 workgroups per CU) in use, in 10 slices of the span
    span = en[ran].max()
    edges = np.linspace(0, span, 11)
    occ = []
    for a, b in zip(edges[:-1], edges[1:]):
        overlap = np.clip(np.minimum(en[ran], b) - np.maximum(st[ran], a), 0, None).sum() / (b - a)
        occ.append(round(float(overlap) / 1280.0, 3))
    out["timeline_us"]["slot_occupancy_by_tenth"] = occ
    np.save(str(ROOT / "gpurun_out" / "gs_wg_timeline.npy"), w)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
