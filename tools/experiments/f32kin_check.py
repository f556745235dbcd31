#!/usr/bin/env python3
"""Experiment (round 5, verdict item 7): the goal workgroups' kinematics in float32 with float64 translations
(make -C omg-planner_amd/csrc BUILD=build_f32kin OUT=libomg_hip_f32kin.so EXTRA=-DOMGX_GS_F32KIN=1 — never the shipped build).

    python tools/experiments/f32kin_check.py costs <lib.so> out.npy      # goal costs of the bench workload's first iteration
    python tools/experiments/f32kin_check.py compare a.npy b.npy          # max relative difference, arg-min flips per scene
    python tools/experiments/f32kin_check.py fuzz <lib.so> <trials> <seed>  # tests/fuzz/fuzz_parity.py on that library (batch kernel only)
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402


def main():
    mode = sys.argv[1]
    if mode == "compare":
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-30)
        flips = int((a.argmin(-1) != b.argmin(-1)).sum())
        srt = np.sort(b, axis=-1)
        print({"goals": int(a.size), "max_rel_cost_diff": float(rel.max()), "mean_rel_cost_diff": float(rel.mean()), "bit_equal_share": float((a == b).mean()),
               "argmin_flips": flips, "scenes": int(a.shape[0]), "smallest_relative_gap_between_best_two_goals": float(((srt[:, 1] - srt[:, 0]) / srt[:, 0]).min())})
        return
    from omg_planner_amd import _lib
    _lib.LIB_PATH = Path(sys.argv[2]).resolve()
    if mode == "costs":
        import torch
        import bench
        from omg_planner_amd.engine import ChompEngine
        cfg, model, batch, start, goals = bench.build_workload(100, 64, 30, 64, 0, False)
        eng = ChompEngine(model, batch, cfg, start, goals, device="cuda:0", ol_alg="MD")
        eng.t = 0
        eng.iterate(0)
        torch.cuda.synchronize()
        np.save(sys.argv[3], eng.goal_cost_total().cpu().numpy())
    elif mode == "fuzz":
        os.environ["OMGX_FUZZ_NO_LATENCY"] = "1"
        sys.path.insert(0, str(ROOT / "tests" / "fuzz"))
        import fuzz_parity
        sys.exit(fuzz_parity.main(int(sys.argv[3]), int(sys.argv[4])))


if __name__ == "__main__":
    main()
