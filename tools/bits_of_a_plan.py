"""SHA-256 of what a plan leaves behind (trajectories, info, learner state, goal indices) for a library variant: run it once per
variant (separate processes) and compare the lines — a change that claims to keep every bit must print the same digests.
    python tools/bits_of_a_plan.py [--lib omg-planner_amd/csrc/libomg_hip_base.so] [--scenes 3] [--goals 64] [--alg MD]"""
import argparse
import copy
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402
from omg_planner_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--scenes", type=int, default=3)
    ap.add_argument("--goals", type=int, default=64)
    ap.add_argument("--alg", default="MD")
    ap.add_argument("--latency", type=int, default=0)
    ap.add_argument("--standoff", type=int, default=0)
    ap.add_argument("--no-poses", action="store_true", help="latency mode without the pose hand-over between the launches")
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = Path(args.lib).resolve()
    from omg_planner_amd.engine import ChompEngine
    if args.no_poses:
        ChompEngine.LAT_HAND_OVER_POSES = False
    cfg, model, batch, start, goals = bench.build_workload(args.scenes, args.goals, 30, 32, 0, False)
    out = {"lib": Path(_lib.LIB_PATH).name}
    for split in (None, False):
        e = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device="cuda:0", ol_alg=args.alg, latency_mode=bool(args.latency))
        e.split_update = split
        e.plan(early_stop=False)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for k in ("traj", "info", "learner_state", "goal_idx", "grad", "cost_traj"):
            h.update(getattr(e, k).cpu().numpy().tobytes())
        out["split" if split is None else "fused"] = h.hexdigest()[:16]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
