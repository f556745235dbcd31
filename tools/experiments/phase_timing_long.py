"""The update launch's marks over the first iterations of a LONG plan (config 5's shape): where 86 us go.  Needs libomg_hip_pt.so.
    python tools/experiments/phase_timing_long.py [scenes] [goals] [waypoints] [objects]"""
import copy, ctypes as C, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench
from omg_planner_amd import _lib
_lib.LIB_PATH = ROOT / "omg-planner_amd" / "csrc" / "libomg_hip_pt.so"
from omg_planner_amd.engine import ChompEngine
S, G, n, obj = [int(x) for x in (sys.argv[1:5] + ["16", "64", "50", "12"][len(sys.argv) - 1:])]
cfg, model, batch, start, goals = bench.build_workload(S, G, n, 64, 0, False, num_objects=obj)
eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device="cuda:0", ol_alg="MD")
eng.select_initial_goal(); eng.pose_hand_over(True)
lib = _lib.lib()
for t in range(0, 30):
    eng.iterate(t); torch.cuda.synchronize()
    if t in (0, 1, 2, 5, 8, 12, 16, 20, 25, 29):
        buf = (C.c_ulonglong * 48)(); lib.omgx_debug_chomp_phase_times(buf, 48)
        tt = np.array(list(buf), dtype=np.float64)
        lb = (C.c_ulonglong * 16)(); lib.omgx_debug_learner_phase_times(lb, 16)
        l = np.array(list(lb), dtype=np.float64)
        print(f"t={t}: step wg phases from 0: {(tt[:9] - tt[0]).astype(int).tolist()} | learner wg {int(tt[27] - tt[26])} cycles: cost vector {int(l[1] - l[0])}, projection {int(l[2] - l[1])}, outer iterations of expert 4: {int(l[8])}")
