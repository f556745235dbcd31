"""CPU tests: oracle goal-set / batch obstacle cost against Cost.batch_obstacle_cost outputs of the
reference (tests/golden/batch_*.npz)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import helpers as H


@pytest.mark.parametrize("case", ["arc_g6_n30", "arc_g5_n7", "arc_attached_g4_n12"])
def test_goalset_cost_matches_reference(case):
    fx = H.load(f"batch_{case}.npz")
    m = H.model_from(fx)
    n = int(fx["n_remaining"])
    G = fx["goals"].shape[0]
    cost, col, pots = orc.goalset_cost(m.blob(), m.points_per_link, H.batch_from(fx), fx["traj_start"][None],
                                       fx["goals"][None], n, float(fx["cfg_dt"]), soften_fingers=int(fx["uncheck"]) == -1,
                                       want_potentials=True)
    ref = fx["potentials"].reshape(G, n, 10, -1)
    # float32 on both sides; the reference differences positions with a float32 matmul (config.py:170)
    np.testing.assert_allclose(pots[0], ref, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(cost[0], fx["goal_cost"], rtol=1e-5, atol=1e-6)
    assert col[0].sum() == fx["collides"].sum()


def test_batch_without_arc_length_matches_reference():
    fx = H.load("batch_noarc_soft_g8.npz")
    m = H.model_from(fx)
    pot, grad, col = orc.fk_sdf(m.blob(), m.points_per_link, H.batch_from(fx), fx["joints"][None], soften_fingers=True)
    np.testing.assert_allclose(pot[0], fx["potentials"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(grad[0], fx["grads"], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(col[0], fx["collides"])
    assert (col[0][:, -2:] == 0).all()


def test_interpolation_matches_reference_joints():
    """multi_interpolate_waypoints(..., 'linear') restated as start + (i+1)/(n+1) * (goal - start)."""
    fx = H.load("batch_arc_g6_n30.npz")
    n = int(fx["n_remaining"])
    t = (np.arange(1, n + 1) / (n + 1.0))[None, :, None]
    mine = fx["traj_start"][None, None] + t * (fx["goals"][:, None] - fx["traj_start"][None, None])
    np.testing.assert_allclose(mine.reshape(-1, 9), fx["joints"], rtol=0, atol=1e-14)
