#!/usr/bin/env python3
"""Can the update launches be hidden behind goal-set launches by STAGGERING the pipeline's parts with events?

The engine's pipeline lets its parts run free on their streams; they fall into lock-step (all goal-set launches at once, then
all update launches with the GPU mostly idle: ~40 of ~226 us per step).  Here the same launches are issued with an explicit
gate — goal-set launch k of the round-robin order may start only when goal-set launch k - `lag` has finished — so that at most
`lag` goal-set launches are in flight and each part's update runs beside the others' goal-set launches.  Variants: update on the
part's own stream or on a high-priority stream of its own.

    python tools/ab_stagger.py [scenes] [parts] [steps]
"""
import copy
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

import bench
from omg_planner_amd.engine import ChompEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dev = torch.device("cuda:0")
cfg, model, batch, start, goals = bench.build_workload(S, 64, 30, 64, 0, False)


def fresh(k):
    eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    eng.pipeline = k
    for _ in range(8):  # schedules measured, prepared calls built
        eng.t = 0
        eng.iterate(0)
    eng.join()
    torch.cuda.synchronize()
    return eng


def baseline(k):
    eng = fresh(k)
    t0 = time.perf_counter()
    for _ in range(STEPS):
        eng.t = 0
        eng.iterate(0)
    eng.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


def staggered(k, lag, prio_update):
    eng = fresh(k)
    parts = eng._parts
    sG = [torch.cuda.Stream(device=dev) for _ in parts]
    sU = [torch.cuda.Stream(device=dev, priority=-1) if prio_update else sG[i] for i in range(len(parts))]
    cur = torch.cuda.current_stream(dev)
    for s_ in set(sG + sU):
        s_.wait_stream(cur)
    ncu = torch.cuda.get_device_properties(dev).multi_processor_count
    g_events = []  # completion events of the goal-set launches in global order
    u_done = [None] * len(parts)
    t0 = time.perf_counter()
    for step in range(STEPS):
        for i, part in enumerate(parts):
            calls = part._hot[1]
            part.t = 1
            prm = part._learner_params()
            if u_done[i] is not None and sU[i] is not sG[i]:
                sG[i].wait_event(u_done[i])  # the part's own previous update (other stream)
            if lag and len(g_events) >= lag:
                sG[i].wait_event(g_events[-lag])
            calls.goalset_layer(prm.start_idx, part._masked, part.schedule, None, sG[i].cuda_stream)
            ev = torch.cuda.Event()
            ev.record(sG[i])
            g_events.append(ev)
            part._schedule()
            if sU[i] is not sG[i]:
                sU[i].wait_event(ev)
            calls.update(prm, part._params(True), 2 * part.S <= ncu, part._next_ticket(), False, sU[i].cuda_stream)
            if sU[i] is not sG[i]:
                u_done[i] = torch.cuda.Event()
                u_done[i].record(sU[i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


print(f"{S} scenes x 64 goals, {STEPS} steps")
for k in (2, K):
    print(f"engine pipeline, {k} parts (free-running)          : {baseline(k):.4f} ms per step", flush=True)
for k in sorted({2, 3, K, 4}):
    for lag in (0, 1, 2, 3):
        if lag > k:
            continue
        for prio in (False, True):
            ms = staggered(k, lag, prio)
            print(f"{k} parts, gate lag {lag} (0 = none), update on {'a high-priority stream' if prio else 'the part stream      '}: {ms:.4f} ms per step", flush=True)
