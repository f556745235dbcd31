"""Experiment: cross-stream dependencies of an iteration through hipStreamWriteValue32 / hipStreamWaitValue32 instead of
events (run on the GPU box).  Prints ms/step for both."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, ".")
import torch

import bench
from omg_planner_amd.engine import ChompEngine

hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
hip.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]


class ValueSync:
    def __init__(self):
        self.p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(self.p), 8, 0x2)
        assert rc == 0, rc
        self.n = 0

    def signal(self, stream):
        self.n += 1
        rc = hip.hipStreamWriteValue32(C.c_void_p(stream.cuda_stream), self.p, self.n, 0)
        assert rc == 0, rc

    def wait(self, stream):
        rc = hip.hipStreamWaitValue32(C.c_void_p(stream.cuda_stream), self.p, self.n, 0, 0xFFFFFFFF)  # Gte
        assert rc == 0, rc


def iterate_value(self, fork: ValueSync, join: ValueSync):
    main = torch.cuda.current_stream(self.device)
    fork.signal(main)
    fork.wait(self.side_stream)
    with torch.cuda.stream(self.side_stream):
        self._layer()
        join.signal(self.side_stream)
    lprm = self.update_goal(defer_update=True)
    self._schedule()
    join.wait(main)
    self._step(True, lprm)


cfg, model, batch, start, goals = bench.build_workload(100, 64, 30, 64, 0, False)
for mode in ("events", "values", "events", "values"):
    eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD")
    fork, join = ValueSync(), ValueSync()

    def step():
        eng.t = 0
        if mode == "events":
            eng.iterate(0)
        else:
            iterate_value(eng, fork, join)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    torch.cuda.synchronize()
    print(mode, (time.perf_counter() - t0) / 50 * 1e3, "ms/step")
