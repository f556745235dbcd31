"""GPU tests of round 6: scenes with more objects than the culling masks have bits, the persistent planner launch (one launch
for K iterations of every scene: omgx_plan_persistent), Learner.reset on the device.  Everything through the C ABI."""
from __future__ import annotations

import copy

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _crowded(num_scenes, num_goals, n, num_objects, grid=16):
    import bench
    return bench.build_workload(num_scenes, num_goals, n, grid, 7, False, num_objects=num_objects)


@pytest.mark.parametrize("latency", [False, True])
def test_more_objects_than_mask_bits(dev, latency):
    """40 obstacle volumes + the table per scene: the row masks have 32 bits and objects >= 31 share the last one
    (omg_goalset_queue.h, cull_row) — the culling gets coarser, the results must not change.  Goal costs, layer outputs and three
    planner iterations against the oracle, which culls nothing."""
    from omg_planner_amd.engine import ChompEngine
    from oracle import oracle as orc
    from oracle.check import engine_vs_oracle
    cfg, model, batch, start, goals = _crowded(2, 8, 30, 40)
    assert int(batch.scene_begin[1] - batch.scene_begin[0]) == 41
    eng = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", latency_mode=latency)
    traj0 = eng.traj.cpu().numpy().copy()
    eng.update_goal(with_layer=True)
    torch.cuda.synchronize()
    P, blob = model.points_per_link, model.blob()
    cost, col = orc.goalset_cost(blob, P, batch, traj0[:, 0], goals, 30, cfg.time_interval)
    np.testing.assert_allclose(eng.goal_cost_total().cpu().numpy(), cost, rtol=1e-5, atol=1e-6)
    pot, pg, cl = orc.fk_sdf(blob, P, batch, traj0)
    d_pot = eng.pot.cpu().numpy()
    assert (d_pot != 0).sum() > 100  # the scene is crowded enough to matter
    np.testing.assert_allclose(d_pot, pot, rtol=0, atol=5e-6)
    np.testing.assert_allclose(eng.pgrad.cpu().numpy(), pg, rtol=0, atol=2e-4)
    assert (eng.col.cpu().numpy() != cl).mean() < 2e-4
    eng2 = ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", latency_mode=latency)
    out = engine_vs_oracle(eng2, batch, [0, 1], steps=3, pin_window=False)
    assert out["ok"] and out["max_traj_err"] < 1e-6, out
