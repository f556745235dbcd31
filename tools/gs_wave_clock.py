"""Per-wave times of a goal workgroup's chain / culling stage (instrumented build, see tools/gs_phase_clock.py):
when each of the four waves entered the stage (after the joints' matrices have been tabulated) and when it finished its part —
waves 0-1 the kinematic chain, waves 2-3 the row culling beside it.  python tools/gs_wave_clock.py [scenes] [goals]"""
import ctypes as C
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

import bench
from omg_planner_amd import _lib
_lib.LIB_PATH = ROOT / "omg-planner_amd" / "csrc" / os.environ.get("OMGX_CLK_LIB", "libomg_hip_clk.so")
from omg_planner_amd.engine import ChompEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
G = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 64, 0, False)
eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD")
lib = _lib.lib()
for _ in range(13):
    eng.t = 0
    eng.iterate(0)
torch.cuda.synchronize()
nwg = ((S + 7) // 8) * 5 * 8 + max(((S + 7) // 8) * G * 8, 0 if eng.schedule is None else int(eng.schedule.numel()))
wg = (C.c_ulonglong * (8 * nwg))()
wv = (C.c_ulonglong * (8 * nwg))()
lib.omgx_debug_gs_wg.argtypes = [C.c_void_p, C.c_int]
lib.omgx_debug_gs_wave.argtypes = [C.c_void_p, C.c_int]
assert lib.omgx_debug_gs_wg(wg, nwg) == 0 and lib.omgx_debug_gs_wave(wv, nwg) == 0
w = np.array(list(wg), dtype=np.uint64).reshape(nwg, 8).astype(np.int64)
v = np.array(list(wv), dtype=np.uint64).reshape(nwg, 8).astype(np.int64)
nlayer = ((S + 7) // 8) * 5 * 8
ran = (w[:, 4] > 0) & (w[:, 0] + 100000 > w[:, 4].max()) & (w[:, 4] > w[:, 0]) & (np.arange(nwg) >= nlayer) & (v[:, 0] > w[:, 0])
us = lambda a: round(float(a[ran].mean()) / 100.0, 2)
out = {"goal_workgroups": int(ran.sum()),
       "entry->after sincos barrier": us(w[:, 1] - w[:, 0]),
       "sincos barrier->stage entered (matrices tabulated + barrier), per wave": [us(v[:, 4 + k] - w[:, 1]) for k in range(2)],
       "stage entered->wave done, per wave (0-1 chain then culling, 2-3 culling)": [us(v[:, k] - v[:, 4 + (k & 1)]) for k in range(4)],
       "stage entered->chain finished (waves 0, 1)": [us(v[:, 6 + k] - v[:, 4 + k]) for k in range(2)],
       "wave done->stage barrier passed (thread 0), per wave": [us(w[:, 2] - v[:, k]) for k in range(4)],
       "sincos barrier->stage barrier": us(w[:, 2] - w[:, 1]),
       "stage barrier->main loop start": us(w[:, 3] - w[:, 2]),
       "main loop (first / last wave out)": [us(w[:, 5] - w[:, 3]), us(w[:, 6] - w[:, 3])],
       "whole": us(w[:, 4] - w[:, 0])}
print(json.dumps(out, indent=1))
