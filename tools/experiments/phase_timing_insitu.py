"""tools/phase_timing.py IN SITU: the update launch's phase marks (scene 0 of whichever part wrote last) while the other pipeline parts' goal-set
launches run beside it, against the same engine run one launch at a time.  Needs libomg_hip_pt.so (-DOMGX_PHASE_TIMING).
    python tools/experiments/phase_timing_insitu.py [scenes] [goals]"""
import copy, ctypes as C, json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import bench
from omg_planner_amd import _lib
_lib.LIB_PATH = ROOT / "omg-planner_amd" / "csrc" / os.environ.get("OMGX_PT_LIB", "libomg_hip_pt.so")
from omg_planner_amd.engine import ChompEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 13
G = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 64, 0, False)
eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=S, device="cuda:0", ol_alg="MD")
eng.pose_hand_over(True)
lib = _lib.lib()

def marks():
    buf = (C.c_ulonglong * 48)(); lib.omgx_debug_chomp_phase_times(buf, 48)
    t = np.array(list(buf), dtype=np.float64)
    lb = (C.c_ulonglong * 16)(); lib.omgx_debug_learner_phase_times(lb, 16)
    l = np.array(list(lb), dtype=np.float64)
    return {"step_phase_starts": (t[:9] - t[0]).astype(int).tolist(), "step_total": int(t[8] - t[0]),
            "extra": {str(k): int(t[k] - t[0]) for k in range(9, 40) if t[k] > 0 and k not in (26, 27)},
            "learner_wg": int(t[27] - t[26]), "learner_start_vs_step_start": int(t[26] - t[0]),
            "learner_cost_vector/projection/expert_cost": [int(l[1] - l[0]), int(l[2] - l[1]), int(l[3] - l[2])]}

def step():
    eng.t = 0
    eng.iterate(0)

for _ in range(30):
    step()
eng.join(); torch.cuda.synchronize()
out = {"scenes": S, "goals": G, "pipeline": eng.pipeline, "in_situ": [], "alone": []}
for rep in range(4):  # free-running pipeline: the marks belong to one of the last update launches
    for _ in range(40 + rep):
        step()
    eng.join(); torch.cuda.synchronize()
    out["in_situ"].append(marks())
for rep in range(4):  # one step at a time: every launch alone on the chip (the parts still overlap inside a step)
    step(); eng.join(); torch.cuda.synchronize()
    out["alone"].append(marks())
print(json.dumps(out))
