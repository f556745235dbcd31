#!/usr/bin/env python3
"""How often each block of k_goalset_queue's main loop runs on the bench workload (per goal workgroup, wave-level counts).

Needs the counting build:  make -C omg-planner_amd/csrc BUILD=build_cnt OUT=libomg_hip_cnt.so EXTRA=-DOMGX_GS_COUNT=1
Run on the GPU box:  python tools/gs_block_counts.py [num_scenes]
"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch

import bench
from omg_planner_amd import _lib
_lib.LIB_PATH = Path(__file__).resolve().parents[1] / "omg-planner_amd" / "csrc" / "libomg_hip_cnt.so"
from omg_planner_amd.engine import ChompEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
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
         "(tile, object) iterations"]
for n, v in zip(names, c):
    print(f"{n:50s} {v:10.1f} per goal workgroup")
