#!/usr/bin/env python3
"""How often each block of k_goalset_queue's main loop runs on the bench workload (per goal workgroup, wave-level counts).

Needs the counting build:  make -C omg-planner_amd/csrc BUILD=build_cnt OUT=libomg_hip_cnt.so EXTRA=-DOMGX_GS_COUNT=1
Run on the GPU box:  python tools/gs_block_counts.py [num_scenes] [--json out.json]

--json: also the triage statistics bench.py's `roofline.pairs` carries — (point, object) pairs tested per launch of all scenes
(every point of every configuration against every enabled object), pairs that survive the row + box tests and are queued for
the exact path, and pairs that contribute a potential or a collision in the end.
"""
import json
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import bench
from omg_planner_amd import _lib
_lib.LIB_PATH = Path(__file__).resolve().parents[1] / "omg-planner_amd" / "csrc" / "libomg_hip_cnt.so"
from omg_planner_amd.engine import ChompEngine

args = [a for a in sys.argv[1:]]
json_out = None
if "--json" in args:
    json_out = args[args.index("--json") + 1]
    del args[args.index("--json"): args.index("--json") + 2]
S = int(args[0]) if args else 100
G = 64
cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 64, 0, False)
eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD")
lib = _lib.lib()
for it in range(3):
    eng.t = 0
    eng.iterate(0)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
lib.omgx_debug_gs_counts(buf, 1)
launches = 4
for it in range(launches):
    eng.t = 0
    eng.iterate(0)
torch.cuda.synchronize()
lib.omgx_debug_gs_counts(buf, 1)
c = [buf[i] / (launches * S * G) for i in range(16)]
names = ["waves", "tiles visited", "tiles entered (a row in reach)", "(tile, object) iterations with a row in reach", "far tests (per link)",
         "far tests with a live lane", "weights computed", "enqueue calls", "issue calls (exact-path batches)", "live lanes enqueued",
         "(tile, object) iterations", "contributing entries", "far tests if tiles were 2 waypoints x 2 links", "enqueue calls if tiles were 2 x 2"]
for n, v in zip(names, c):
    print(f"{n:50s} {v:10.1f} per goal workgroup")
if json_out:
    n, P = 30, model.points_per_link
    enabled = int((batch.objects["disabled"] <= 0).sum())  # over all scenes
    tested = G * n * 10 * P * enabled  # per launch of all S scenes: every point of every configuration x every enabled object of its scene
    per_launch = lambda k: buf[k] / launches
    out = {"scenes": S, "goals": G, "per_goal_workgroup": {nm: v for nm, v in zip(names, c)},
           "pairs": {"tested": tested, "box_survivors": per_launch(9), "contributing": per_launch(11),
                     "survivors_per_contributor": per_launch(9) / max(per_launch(11), 1.0),
                     "exact_path_batches": per_launch(8)}}
    Path(json_out).write_text(json.dumps(out, indent=1) + "\n")
