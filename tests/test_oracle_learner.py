"""CPU tests: the oracle's Learner.update_goal restatement (cost-vector tail + FTL / FTC / Exp / MD) against
fixtures recorded from the reference's own Learner (tests/golden/learner_*.npz)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import helpers as H

CASES = ["FTL_0", "FTC_0", "Exp_0", "MD_0", "MD_1", "FTC_0_close", "MD_0_close", "MD_0_reset", "Exp_0_reset"]  # _reset: Learner.reset(traj) on the way


def learner_params(fx, t):
    p = orc.LearnerParams()
    G = fx["goal_set"].shape[0]
    n = fx["traj"].shape[0]
    p.alg = orc.ALG[str(fx["alg"])]
    p.num_goals, p.n_waypoints = G, n
    p.start_idx = min(int((t / int(fx["optim_steps"])) * n), n - 1)
    p.use_standoff = int(fx["cfg_use_standoff"])
    p.constraint_num = fx["reach_grasps"].shape[1] if p.use_standoff else 1
    p.normalize_cost = int(fx.get("cfg_normalize_cost", 1))
    p.base_obstacle_weight = float(fx.get("cfg_base_obstacle_weight", 1.0))
    p.smooth_weight = float(fx.get("cfg_smoothness_base_weight", 0.1)) * float(fx["dist_eps"])
    p.eta = float(fx["eta"])
    return p


def run_oracle_sequence(fx, goal_cost_fn):
    """Replays the fixture's update_goal calls; goal_cost_fn(step, params) -> [1,G] float32 obstacle sums."""
    G = fx["goal_set"].shape[0]
    state = orc.learner_state_init(1, G)
    # Learner.__init__ runs one cost_vector at t = 0 for the initial goal (online_learner.py:97-102); it does not
    # touch the learner state, so the replay starts from the initial state
    out = []
    reset_at = int(fx["reset_at"]) if "reset_at" in fx else -1
    for k in range(fx["trajs"].shape[0]):
        if k == reset_at:  # Learner.reset (omg/online_learner.py:251-263): p, sum_costs and t start again, experts / q stay
            state[0, :G] = 0.0
            state[0, G:2 * G] = 1.0 / G
        prm = learner_params(fx, float(fx["t_of_step"][k]) if "t_of_step" in fx else k + 1)
        gc = goal_cost_fn(k, prm)
        idx, end, rows, gp, cv = orc.goal_update(prm, fx["trajs"][k][None], fx["goal_set"][None], fx["reach_grasps"][None], gc, state)
        out.append((idx[0], cv[0].copy(), state[0, G:2 * G].copy(), state[0, 7 * G:7 * G + 5].copy(), end[0], rows[0], gp[0]))
    return out, state


@pytest.mark.parametrize("case", CASES)
def test_goal_update_matches_reference_learner(case):
    fx = H.load(f"learner_{case}.npz")
    m = H.model_from(fx)
    batch = H.batch_from(fx)
    G = fx["goal_set"].shape[0]
    cv_goals = fx["reach_grasps"][:, -1, :] if int(fx["cfg_use_standoff"]) else fx["goal_set"]

    def goal_cost(k, prm):
        n_rem = prm.n_waypoints - prm.start_idx
        cost, _ = orc.goalset_cost(m.blob(), m.points_per_link, batch, fx["trajs"][k][prm.start_idx][None], cv_goals[None],
                                   n_rem, float(fx["cfg_dt"]))
        return cost

    out, state = run_oracle_sequence(fx, goal_cost)
    for k, (idx, cv, p, q, end, rows, gp) in enumerate(out):
        np.testing.assert_allclose(cv, fx["cost_vectors"][k], rtol=2e-5, atol=1e-7, err_msg=f"cost vector step {k}")
        np.testing.assert_allclose(p, fx["p"][k], rtol=1e-4, atol=1e-6, err_msg=f"p step {k}")
        assert idx == fx["goal_idx"][k]
        np.testing.assert_array_equal(end, fx["goal_set"][idx])
        np.testing.assert_array_equal(gp, fx["goal_set"][idx])
        np.testing.assert_array_equal(rows, fx["reach_grasps"][idx] if int(fx["cfg_use_standoff"]) else fx["goal_set"][idx][None])
        if str(fx["alg"]) == "MD":
            np.testing.assert_allclose(q, fx["q"][k], rtol=1e-4, atol=1e-7, err_msg=f"q step {k}")
    if str(fx["alg"]) in ("FTL", "Exp"):
        np.testing.assert_allclose(state[0, :G], fx["sum_costs"], rtol=2e-5)
    if str(fx["alg"]) == "MD":
        np.testing.assert_allclose(state[0, 2 * G:7 * G].reshape(5, G), fx["experts_p"], rtol=1e-4, atol=1e-6)


def test_proj_picks_closest_goal():
    fx = H.load("learner_FTL_0.npz")
    prm = learner_params(fx, 1)
    prm.alg = orc.ALG["Proj"]
    G = fx["goal_set"].shape[0]
    state = orc.learner_state_init(1, G)
    idx, end, rows, gp, _ = orc.goal_update(prm, fx["trajs"][0][None], fx["goal_set"][None], None, np.zeros((1, G), np.float32), state)
    d = np.linalg.norm(fx["trajs"][0][-1][None] - fx["goal_set"], axis=-1)
    assert idx[0] == int(np.argmin(d))
    np.testing.assert_array_equal(end[0], fx["goal_set"][idx[0]])


@pytest.mark.parametrize("alg", ["FTL", "FTC", "Exp"])
def test_degenerate_cost_vector_selects_like_numpy(alg):
    """A zero cost vector normalises to 0/0 = NaN (online_learner.py:153-156).  np.argmin / np.argmax then return the
    FIRST NaN, i.e. goal 0 — the index must stay valid (found by tests/fuzz/fuzz_parity.py: one goal, one-waypoint window)."""
    fx = H.load("learner_FTL_0.npz")
    prm = learner_params(fx, 1)
    prm.alg = orc.ALG[alg]
    G = fx["goal_set"].shape[0]
    traj = fx["trajs"][0][None].copy()
    goals = np.repeat(traj[:, prm.start_idx][:, None, :], G, axis=1)   # every goal == traj_start: smoothness proxy 0
    state = orc.learner_state_init(1, G)
    idx, end, rows, gp, cv = orc.goal_update(prm, traj, goals, None, np.zeros((1, G), np.float32), state)
    assert np.isnan(cv).all()
    assert idx[0] == 0 == int(np.argmin(cv[0])) == int(np.argmax(cv[0]))
    np.testing.assert_array_equal(end[0], goals[0, 0])


@pytest.mark.parametrize("case", CASES)
def test_initial_goal_pick_matches_reference_learner_init(case):
    """Learner.__init__ (online_learner.py:96-102): one cost_vector at t = 0 on the initial trajectory, goal = argmin."""
    fx = H.load(f"learner_{case}.npz")
    m = H.model_from(fx)
    cv_goals = fx["reach_grasps"][:, -1, :] if int(fx["cfg_use_standoff"]) else fx["goal_set"]
    prm = learner_params(fx, 0)
    prm.alg = orc.ALG["FTC"]  # argmin of the cost vector, no state
    n = prm.n_waypoints
    cost, _ = orc.goalset_cost(m.blob(), m.points_per_link, H.batch_from(fx), fx["traj"][0][None], cv_goals[None], n, float(fx["cfg_dt"]))
    state = orc.learner_state_init(1, fx["goal_set"].shape[0])
    idx, end, _, _, _ = orc.goal_update(prm, fx["traj"][None], fx["goal_set"][None], fx["reach_grasps"][None], cost, state)
    assert int(idx[0]) == int(fx["init_goal_idx"])
    np.testing.assert_array_equal(end[0], fx["goal_set"][int(fx["init_goal_idx"])])
