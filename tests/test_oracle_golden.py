"""CPU tests: the oracle (oracle/omg_oracle.c) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  These pin rows 7-23 of SURVEY.md §8a; the SDF op (rows 1-3) enters
them as recorded inputs and is checked separately in test_oracle_sdf.py."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import helpers as H

COST_CASES = ["topk1000", "topk300", "clean", "fixed_end", "soft_finger", "finger_n50", "attached", "short_n5"]
OPT_CASES = ["standoff_20", "nostandoff_20", "fixed_end_5", "limits_5", "n50_dt006_5"]


def test_struct_sizes():
    lib = orc.lib()
    assert lib.orc_sizeof_object() == 184


def test_fk_matches_reference():
    fx = H.load("fk.npz")
    m = H.model_from(fx)
    pose, org, ax = orc.fk(m.blob(), fx["joints"])
    np.testing.assert_allclose(pose, fx["poses"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(org, fx["joint_origins"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ax, fx["joint_axis"], rtol=0, atol=1e-12)


def test_smoothness_matrices_match_reference():
    fx = H.load("matrices.npz")
    for tag in sorted({k.rsplit("_", 1)[0] for k in fx}):
        n = int(tag.split("_")[0][1:])
        gsp = int(tag.split("_")[1][1:])
        dt = float(tag.split("dt")[1])
        D, A, Ainv = orc.smooth_matrices(n, dt, gsp)
        np.testing.assert_allclose(D, fx[tag + "_D1"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(A, fx[tag + "_A"], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(Ainv, fx[tag + "_Ainv"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("case", COST_CASES)
def test_sdf_layer_chain_matches_reference(case):
    """FK -> points -> layer: the potentials the reference fed to compute_collision_loss."""
    fx = H.load(f"cost_{case}.npz")
    m = H.model_from(fx)
    n = fx["xi"].shape[0]
    pot, grad, col = orc.fk_sdf(m.blob(), m.points_per_link, H.batch_from(fx), fx["xi"][None],
                                soften_fingers=int(fx["cfg_uncheck"]) == -1)
    # same op on both sides; the only difference is the last-ulp of the float64 FK feeding float32 points
    np.testing.assert_allclose(pot[0], fx["potentials"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(grad[0], fx["potential_grads"], rtol=0, atol=2e-4)
    assert col.sum() == fx["collide_sum"]
    assert (pot[0] == fx["potentials"]).mean() > 0.99


@pytest.mark.parametrize("case", COST_CASES)
def test_total_loss_matches_reference(case):
    fx = H.load(f"cost_{case}.npz")
    m = H.model_from(fx)
    n, P = fx["xi"].shape[0], m.points_per_link
    _, _, col = orc.fk_sdf(m.blob(), P, H.batch_from(fx), fx["xi"][None], soften_fingers=int(fx["cfg_uncheck"]) == -1)
    prm = H.params_from(fx, orc.ChompParams, n, P, 0, float(fx["cfg_obstacle_weight"]), float(fx["cfg_smoothness_weight"]))
    c = prm.constraint_num
    goal = np.tile(fx["end"], (1, c, 1))
    traj, grad, cost_traj, info = orc.chomp_optimize(
        m.blob(), prm, fx["xi"][None], fx["start"][None], fx["end"][None], goal, fx["goal_point"][None],
        fx["potentials"][None], fx["potential_grads"][None], col)
    np.testing.assert_array_equal(traj[0], fx["xi"])  # info_only
    np.testing.assert_allclose(grad[0], fx["total_grad"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(grad[0], fx["info_gradient"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(cost_traj[0], fx["info_cost_traj"], rtol=1e-9, atol=1e-9)
    I = H.INFO_IDX
    for key in ["cost", "obs", "smooth", "weighted_obs", "weighted_smooth", "weighted_obs_grad", "weighted_smooth_grad",
                "grad", "collide", "reach", "standoff_idx", "terminate", "failure_terminate", "execute"]:
        np.testing.assert_allclose(info[0, I[key]], fx["info_" + key], rtol=1e-9, atol=1e-9, err_msg=key)
    np.testing.assert_allclose(info[0, I["cost"]], fx["total_cost"], rtol=1e-9)


def _opt_step_inputs(fx, k):
    use_standoff = int(fx["cfg_use_standoff"])
    gi = int(fx["goal_idx"])
    goal = fx["reach_grasps"][gi] if use_standoff else fx["goal_set"][gi][None]
    return goal[None], fx["goal_set"][gi][None]


@pytest.mark.parametrize("case", OPT_CASES)
def test_optimizer_steps_match_reference(case):
    """Optimizer.optimize step by step: teacher-forced (each step restarts from the reference's
    iterate: 1e-9) and free-running over the whole sequence (1e-6)."""
    fx = H.load(f"opt_{case}.npz")
    m = H.model_from(fx)
    hist = fx["traj_history"]
    steps, n = hist.shape[0] - 1, hist.shape[1]
    P = m.points_per_link
    batch = H.batch_from(fx)
    I = H.INFO_IDX
    free = hist[0].copy()
    for k in range(steps + 1):
        w_obs, w_sm, eta = fx["schedule"][k]
        # force_update=False (cfg_force_update 0): a terminated trajectory is returned untouched -> do_update 2
        do_update = (1 if int(fx.get("cfg_force_update", 1)) else 2) if k < steps else 0
        prm = H.params_from(fx, orc.ChompParams, n, P, do_update, w_obs, w_sm, eta, int(fx["cfg_reach_tail_length"]))
        goal, goal_point = _opt_step_inputs(fx, k)
        for mode, x0 in (("forced", hist[k]), ("free", free)):
            pot, pg, col = orc.fk_sdf(m.blob(), P, batch, x0[None])
            traj, grad, _, info = orc.chomp_optimize(m.blob(), prm, x0[None], fx["start"][None], fx["end"][None], goal,
                                                     goal_point, pot[0][None], pg[0][None], col[0][None])
            tol = 1e-9 if mode == "forced" else 1e-6
            if mode == "forced":
                np.testing.assert_allclose(grad[0], fx["info_gradient"][k], rtol=1e-7, atol=1e-7, err_msg=f"step {k}")
                for key in ["cost", "obs", "smooth", "collide", "reach", "terminate", "violate_limit", "execute",
                            "failure_terminate"]:
                    np.testing.assert_allclose(info[0, I[key]], fx["info_" + key][k], rtol=1e-7, atol=1e-7,
                                               err_msg=f"{key} step {k}")
            if do_update:
                np.testing.assert_allclose(traj[0], hist[k + 1], rtol=0, atol=tol, err_msg=f"{mode} step {k}")
            if mode == "free":
                free = traj[0]
    if case == "limits_5":
        assert fx["info_violate_limit"].max() >= 0  # fixture exercised handle_joint_limit


def test_optimize_without_force_update_leaves_a_terminated_trajectory_alone():
    """Optimizer.optimize(traj) with force_update=False (optimizer.py:126-127): `if info["terminate"] and not force_update:
    return` — do_update = 2.  A collision-free straight trajectory that ends on its goal terminates; do_update = 1 moves it."""
    fx = H.load("cost_topk1000.npz")
    m = H.model_from(fx)
    n, P = 6, m.points_per_link
    start = fx["start"]
    xi = start[None] + np.linspace(0, 1e-3, n)[:, None] * np.ones(9)[None] * np.array([1] * 7 + [0, 0])
    end = xi[-1].copy()
    zeros = np.zeros((1, n, 10, P), np.float32)
    out = {}
    for mode in (1, 2):
        prm = H.params_from(fx, orc.ChompParams, n, P, mode, 1.0, 0.1)
        prm.goal_set_proj, prm.use_standoff, prm.constraint_num, prm.pre_terminate = 1, 0, 1, 1
        prm.terminate_smooth_loss = 1e9
        traj, _, _, info = orc.chomp_optimize(m.blob(), prm, xi[None], start[None], end[None], end[None, None], end[None], zeros,
                                              np.zeros((1, n, 10, P, 3), np.float32), zeros)
        assert info[0, H.INFO_IDX["terminate"]] == 1.0
        out[mode] = traj[0]
    np.testing.assert_array_equal(out[2], xi)
    assert np.abs(out[1] - xi).max() > 0


def test_division_by_constant_through_fma_equals_the_quotient():
    """omg_device.h: div_by_const — the kernels' rad -> deg -> rad round trip (omg/util.py:194-202, robot_pykdl.py:164) divides by
    pi and by 180 with a multiplication and one FMA correction step instead of the IEEE division sequence; the two agree on
    every sampled argument."""
    import ctypes as C
    import math
    lib = orc.lib()
    lib.orc_div_by_const_mismatches.restype = C.c_int64
    lib.orc_div_by_const_mismatches.argtypes = [C.c_double, C.c_int64, C.c_uint64]
    for c in (math.pi, 180.0):
        assert lib.orc_div_by_const_mismatches(c, 20_000_000, 12345) == 0
