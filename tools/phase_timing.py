"""Phase breakdown of k_chomp_optimize / k_goal_update (workgroup 0) on the bench workload.

Needs a debug build of the library:  make -C omg-planner_amd/csrc -B CXXFLAGS='-O3 -std=c++17 -fPIC -ffp-contract=off -DOMGX_PHASE_TIMING'
(rebuild without the flag afterwards).  Run on the GPU box:  python tools/phase_timing.py
"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

import bench
from omg_planner_amd import _lib
from pathlib import Path
_lib.LIB_PATH = Path(__file__).resolve().parents[1] / "omg-planner_amd" / "csrc" / "libomg_hip_pt.so"
from omg_planner_amd.engine import ChompEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg, model, batch, start, goals = bench.build_workload(S, 64, 30, 64, 0, True)
eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD")
lib = _lib.lib()
for it in range(6):
    eng.t = 0
    eng.iterate(0)
torch.cuda.synchronize()
for name, fn in (("k_chomp_optimize", "omgx_debug_chomp_phase_times"),):
    buf = (C.c_ulonglong * 32)()
    rc = getattr(lib, fn)(buf, 32)
    t = np.array(list(buf), dtype=np.float64)
    print(name, "rc", rc)
    if name == "k_chomp_optimize":
        d = np.diff(t[:9])
        print("  phase clocks (phases 0..7):", d.astype(int).tolist(), "total", int(t[8] - t[0]))
        print("  extra marks relative to start:", {k: int(t[k] - t[0]) for k in range(16, 30) if t[k] > 0})
    else:
        d = np.diff(t[:5])
        print("  clocks [cost-vector, projection (wave 0), wait for other experts, mixture]:", d.astype(int).tolist(), "total", int(t[4] - t[0]))
        print("  per expert (outer iterations, inner bisection steps):", [(int(t[8 + 2 * w]), int(t[9 + 2 * w])) for w in range(5)])
