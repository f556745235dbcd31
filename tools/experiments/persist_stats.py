"""Where the persistent launch's time goes (measurement build -DOMGX_PERSIST_STATS: make -C omg-planner_amd/csrc BUILD=build_pstats
OUT=libomg_hip_pstats.so EXTRA=-DOMGX_PERSIST_STATS=1): per launch the summed durations of the items, of the updates and of the claims.
    python tools/experiments/persist_stats.py [--scenes 100 --goals 64 --steps 20]"""
import argparse
import copy
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from omg_planner_amd import _lib  # noqa: E402

_lib.LIB_PATH = ROOT / "omg-planner_amd" / "csrc" / "libomg_hip_pstats.so"
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd.engine import ChompEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=100)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--waypoints", type=int, default=30)
    ap.add_argument("--objects", type=int, default=4)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--max-wg", type=int, default=0)
    ap.add_argument("--update-cus", type=int, default=-1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg, model, batch, start, goals = bench.build_workload(args.scenes, args.goals, args.waypoints, 64, 0, False, num_objects=args.objects, device=dev)
    eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    eng.pose_hand_over(True)
    snap = eng.snapshot()
    lib = _lib.lib()
    lib.omgx_debug_persist_stats.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_ulonglong), C.c_void_p]
    out = []
    for rep in range(3):
        eng.restore(snap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_persistent([0] * args.steps, pin_window=True, max_workgroups=args.max_wg, update_cus=args.update_cus)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        st = (C.c_ulonglong * 8)()
        lib.omgx_debug_persist_stats(C.c_void_p(eng._persist_ws.data_ptr()), args.scenes, st, None)
        spins, item_t, upd_t, items, upds, claim_t = (int(st[i]) for i in range(6))
        out.append({"ms_per_step": ms / args.steps, "items": items, "updates": upds, "item_us": item_t / max(items, 1) / 100.0,
                    "update_us": upd_t / max(upds, 1) / 100.0, "claim_us": claim_t / max(items, 1) / 100.0, "spins_per_claim": spins / max(items, 1),
                    "status": eng.persistent_status()})
    print(json.dumps({"shape": [args.scenes, args.goals, args.waypoints, args.objects], "steps": args.steps, "runs": out}))


if __name__ == "__main__":
    main()
