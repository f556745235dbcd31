"""Latency mode: per-workgroup timeline of one goal-set launch (instrumented library, see tools/gs_phase_clock.py).
    python tools/lat_phase_clock.py [scenes] [goals] [window_start]"""
import ctypes as C
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd import _lib, ops  # noqa: E402

_lib.LIB_PATH = ROOT / "omg-planner_amd" / "csrc" / os.environ.get("OMGX_CLK_LIB", "libomg_hip_clk.so")
from omg_planner_amd.engine import ChompEngine  # noqa: E402

import os as _os
if _os.environ.get("OMGX_LAT_PARTS"):
    ChompEngine.LAT_GOAL_PARTS = int(_os.environ["OMGX_LAT_PARTS"])  # experiments


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    t = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 64, 0, False)
    eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD", latency_mode=True)
    lib = _lib.lib()
    for _ in range(8):
        eng.t = t
        eng.iterate(t)
    torch.cuda.synchronize()
    n_rem = 30 - min(int((t / cfg.optim_steps) * 30), 29)
    NP = ops.goalset_parts(n_rem, eng.LAT_GOAL_PARTS)
    LP = eng.LAT_LAYER_LINK_GROUPS * ((30 + eng.LAT_LAYER_BLOCK - 1) // eng.LAT_LAYER_BLOCK)
    if t >= cfg.optim_steps:
        G = 0
    nwg = S * (LP + G * NP)
    wg = (C.c_ulonglong * (8 * nwg))()
    lib.omgx_debug_gs_wg.argtypes = [C.c_void_p, C.c_int]
    assert lib.omgx_debug_gs_wg(wg, nwg) == 0
    w = np.array(list(wg), dtype=np.uint64).reshape(nwg, 8)
    t0 = w[:, 0].min()
    st, t_sc, t_ch, pro, en = [(w[:, k].astype(np.int64) - int(t0)) / 100.0 for k in range(5)]
    first, last = [(w[:, k].astype(np.int64) - int(t0)) / 100.0 for k in (5, 6)]
    is_layer = np.arange(nwg) < S * LP
    g = ~is_layer
    out = {"scenes": S, "goals": G, "parts": NP, "workgroups": nwg, "kernel_span_us": float(en.max()), "last_start_us": float(st.max()),
           "layer_wg sincos/chain+cull mean": [float(x[is_layer].mean()) for x in (t_sc - st, t_ch - t_sc)] if False else None,
           "layer_wg_life all": [round(float(v), 2) for v in (en - st)[is_layer]]}
    if G == 0:
        print(json.dumps(out)); return
    out.update({
           "goal_wg_life mean/max": [float((en - st)[g].mean()), float((en - st)[g].max())],
           "goal_wg sincos/chain/cull mean": [float(x[g].mean()) for x in (t_sc - st, t_ch - t_sc, pro - t_ch)],
           "goal_wg main loop first/last wave mean": [float((first - pro)[g].mean()), float((last - pro)[g].mean())],
           "goal_wg main loop last wave max": float((last - pro)[g].max()),
           "goal_wg epilogue mean": float((en - last)[g].mean()),
           "layer_wg_life mean/max": [float((en - st)[is_layer].mean()), float((en - st)[is_layer].max())],
           "layer_end_max": float(en[is_layer].max()), "goal_end_max": float(en[g].max())})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
