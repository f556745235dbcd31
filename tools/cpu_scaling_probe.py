#!/usr/bin/env python3
"""How many host cores does the oracle's OpenMP loop really get on this box?  Times orc.goalset_cost (embarrassingly parallel
over (scene, goal) items, schedule(dynamic, 1)) for a fixed batch with 1 .. all threads.  Feeds bench.py's choice of the thread
count it reports as `cpu_baseline`."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np

import bench
from oracle import oracle as orc

print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
S, G, n = 8, 64, 30
cfg, model, batch, start, goals = bench.build_workload(S, G, n, 64, 0, True)
blob, P = model.blob(), model.points_per_link
for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if th > (os.cpu_count() or 1):
        break
    orc.set_threads(th)
    orc.goalset_cost(blob, P, batch, start, goals[:, :8], n, cfg.time_interval)  # warm
    t0 = time.perf_counter()
    orc.goalset_cost(blob, P, batch, start, goals, n, cfg.time_interval)
    dt = time.perf_counter() - t0
    print(f"threads {th:4d}: {dt * 1e3:9.1f} ms for {S * G} (scene, goal) items = {S * G / dt:9.1f} items/s")
