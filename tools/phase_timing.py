"""Phase breakdown of k_update_optimize_split (scene 0: its learner workgroup and its step workgroup) on the bench workload.

Needs a debug build of the library (kept out of the product build):
    make -C omg-planner_amd/csrc BUILD=build_pt OUT=libomg_hip_pt.so EXTRA=-DOMGX_PHASE_TIMING
Run on the GPU box:  python tools/phase_timing.py [num_scenes]
Clocks are __builtin_readcyclecounter cycles (s_memtime: the shader clock, ~2.2 GHz under this load).
"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch

import bench
from omg_planner_amd import _lib
import os
_lib.LIB_PATH = Path(__file__).resolve().parents[1] / "omg-planner_amd" / "csrc" / os.environ.get("OMGX_PT_LIB", "libomg_hip_pt.so")
from omg_planner_amd.engine import ChompEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg, model, batch, start, goals = bench.build_workload(S, 64, 30, 64, 0, True)
eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD")
if os.environ.get("OMGX_PT_POSES", "1") != "0":  # as inside plan() and bench.py: the launches hand the link poses to each other
    eng.pose_hand_over(True)
lib = _lib.lib()
for it in range(12):
    eng.iterate(it)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 48)()
    lib.omgx_debug_chomp_phase_times(buf, 48)
    t = np.array(list(buf), dtype=np.float64)
    lb = (C.c_ulonglong * 16)()
    lib.omgx_debug_learner_phase_times(lb, 16)
    l = np.array(list(lb), dtype=np.float64)
    d = np.diff(t[:9])
    print("   phase starts relative to phase 0:", (t[:9] - t[0]).astype(int).tolist())
    print(f"iteration {it}: step workgroup phases 0..7 [shader-clock cycles, ~2.2 GHz]:", d.astype(int).tolist(), "total", int(t[8] - t[0]))
    print("   extra marks relative to phase 0:", {k: int(t[k] - t[0]) for k in list(range(9, 16)) + list(range(16, 40)) if t[k] > 0 and k not in (26, 27)})
    print("   learner wave 4 [shader-clock cycles, ~2.2 GHz]: cost vector", int(l[1] - l[0]), "projection", int(l[2] - l[1]), "expert cost", int(l[3] - l[2]),
          "| mixture wave (from its expert done to the goal): ", int(l[5] - l[4]), "| whole learner workgroup", int(t[27] - t[26]), "| outer iterations of expert 4:", int(l[8]))
