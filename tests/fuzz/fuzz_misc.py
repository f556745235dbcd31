"""Randomised differential test (GPU box vs CPU oracle) of the two remaining entry points: omgx_point_cloud_sdf (bit-exact
nearest-point distance grids for random clouds / resolutions / margins) and omgx_forward_kinematics (poses, joint origins
and axes of random configurations incl. far outside the joint limits, 1e-12).

    python tests/fuzz/fuzz_misc.py [trials] [seed]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from omg_planner_amd import ops, robot as rb, scenes as sc
from oracle import oracle as orc


def main(trials=None, seed=None):
    trials = int(trials if trials is not None else (sys.argv[1] if len(sys.argv) > 1 else 50))
    rng = np.random.RandomState(int(seed if seed is not None else (sys.argv[2] if len(sys.argv) > 2 else 0)))
    dev = torch.device("cuda:0")
    bad, t0, voxels = 0, time.time(), 0
    for k in range(trials):
        errs = []
        try:
            # ---- point cloud -> SDF
            N = int(rng.choice([1, 2, 17, 300, 4096]))
            pts = rng.normal(0, rng.uniform(0.01, 0.3), (N, 3)) + rng.uniform(-1, 1, 3)
            if rng.rand() < 0.2:
                pts[: N // 2] = pts[0]  # duplicates
            res, margin = float(rng.choice([0.02, 0.05, 0.013])), float(rng.choice([0.24, 0.05, 0.1]))
            grid, origin, r = ops.point_cloud_sdf(torch.as_tensor(pts, device=dev), res, margin)
            ref = sc.point_cloud_sdf(pts, res, margin)
            voxels += grid.numel()
            if tuple(grid.shape) != ref.data.shape or not np.array_equal(grid.cpu().numpy().view(np.int32), ref.data.view(np.int32)):
                errs.append("point cloud grid differs")
            if not np.array_equal(origin, ref.origin):
                errs.append("point cloud origin differs")
            # ---- forward kinematics
            m = rb.PandaModel(points_per_link=int(rng.choice([1, 15, 16])), seed=int(rng.randint(0, 99)))
            B = int(rng.choice([1, 63, 64, 65, 1000]))
            q = rng.uniform(-7, 7, (B, 9)) if rng.rand() < 0.5 else rng.uniform(m.joint_lower_limit[0], m.joint_upper_limit[0], (B, 9))
            poses, org, ax = ops.forward_kinematics(ops.robot_blob(m, dev), m.points_per_link, torch.as_tensor(q, device=dev))
            rp, ro, ra = orc.fk(m.blob(), q)
            for nm, a, b in (("poses", poses, rp), ("origins", org, ro), ("axes", ax, ra)):
                if not np.allclose(a.cpu().numpy(), b, rtol=0, atol=1e-12):
                    errs.append(f"fk {nm} differ by {np.abs(a.cpu().numpy() - b).max():.2e}")
        except Exception as e:  # noqa: BLE001
            errs.append(f"exception {type(e).__name__}: {e}")
        if errs:
            bad += 1
            print(f"trial {k}: FAIL " + "; ".join(errs), flush=True)
    print(f"{trials - bad}/{trials} trials agree; {voxels} point-cloud voxels compared bit for bit; {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
