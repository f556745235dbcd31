"""CPU test of bench.py's host-side pieces: workload construction and the cpu_baseline leg (the oracle timed on a
bounded sample); the timed GPU region itself needs an MI355X."""
import numpy as np


def test_workload_and_cpu_baseline():
    import bench
    cfg, model, batch, start, goals = bench.build_workload(3, 4, 30, 16, seed0=0, share_grids=False)
    assert batch.num_scenes == 3 and goals.shape == (3, 4, 9) and start.shape == (3, 9)
    assert len(batch.objects) == 15 and batch.objects["grid_offset"].max() < batch.pool.size
    # private volumes: no two records share a grid
    assert len(set(batch.objects["grid_offset"].tolist())) == 15
    shared = bench.build_workload(3, 4, 30, 16, seed0=0, share_grids=True)[2]
    assert shared.pool.size < batch.pool.size
    # goals are grasp-like: hand within ~0.2 m of the target
    from omg_planner_amd import scenes as sc
    pos, _ = sc.hand_pose(model, goals[0])
    assert np.linalg.norm(pos - sc.make_tabletop_scene(0, grid=16).objects[0].pose_mat[:3, 3], axis=1).max() < 0.45
    out = bench.cpu_baseline(cfg, model, batch, start, goals, 30, budget_s=0.5)
    assert out["kind"] == "port" and out["unit"] == "iterations/s" and out["value"] > 0 and out["cores"] >= 1
