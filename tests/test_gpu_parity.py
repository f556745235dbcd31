"""GPU parity tests (run with -m gpu on an MI355X): the HIP kernels, called through the C ABI of
include/omg_hip.h, against (a) the CPU oracle on the same seeded inputs and (b) the golden fixtures
produced by the reference itself.

Tolerances: the float32 SDF arithmetic is bit-exact against the oracle; float64 stages are compared
at 1e-9 (one step) / 1e-6 (free-running sequences); north_star's bar is 1e-4 on trajectory states
and cost values."""
import ctypes as C

import numpy as np
from pathlib import Path
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    from omg_planner_amd import _lib
    assert _lib.device_arch().startswith("gfx950"), _lib.device_arch()
    return torch.device("cuda:0")


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev)


def _padded_inputs(fx):
    poses, eps, pad, clr, dis = H.layer_params_from(fx)
    return poses, fx["sdf"], fx["limits"], eps, pad, clr, dis


# ------------------------------------------------------------------------------------------------
# (1) omgx_sdf_loss_forward
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["cost_topk300.npz", "cost_soft_finger.npz", "cost_attached.npz"])
def test_sdf_loss_forward_bit_exact_vs_oracle(dev, case):
    from omg_planner_amd import ops
    from oracle import oracle as orc
    fx = H.load(case)
    poses, sdf, lim, eps, pad, clr, dis = _padded_inputs(fx)
    rng = np.random.RandomState(0)
    N = 200_000
    pts = rng.uniform([-0.4, -0.8, -0.3], [1.2, 0.8, 1.2], size=(N, 3)).astype(np.float32)
    # edge cases: far away, exactly on grid faces, huge, non-finite
    pts[:8] = [[1e9, 0, 0], [-1e9, 0, 0], [3e9, 1, 1], [np.inf, 0, 0], [0, -np.inf, 0], [1e30, 1e30, 1e30],
               [0.5, 0.0, 0.02], [-0.3, -0.3, -0.3]]
    ref = orc.sdf_loss_forward(poses, sdf, lim, pts, eps, pad, clr, dis)
    got = ops.sdf_loss_forward(*[_t(a, dev) for a in (poses, sdf, lim, pts, eps, pad, clr, dis)])
    torch.cuda.synchronize()
    for name, r, g in zip(("potentials", "grads", "collides"), ref, got):
        np.testing.assert_array_equal(g.cpu().numpy(), r, err_msg=name)
    assert (ref[0] > 0).mean() > 0.05 and ref[2].sum() > 100  # the case exercises both hinge branches


def test_sdf_loss_forward_empty_and_disabled(dev):
    from omg_planner_amd import ops
    fx = H.load("cost_topk300.npz")
    poses, sdf, lim, eps, pad, clr, dis = _padded_inputs(fx)
    args = [_t(a, dev) for a in (poses, sdf, lim)]
    tail = [_t(a, dev) for a in (eps, pad, clr)]
    empty = torch.zeros((0, 3), dtype=torch.float32, device=dev)
    out = ops.sdf_loss_forward(*args, empty, *tail, _t(dis, dev))
    assert out[0].shape == (0,) and out[1].shape == (0, 3)
    pts = _t(np.random.RandomState(1).uniform(0, 0.8, size=(1000, 3)).astype(np.float32), dev)
    out = ops.sdf_loss_forward(*args, pts, *tail, torch.ones_like(_t(dis, dev)))
    assert float(out[0].abs().sum()) == 0 and float(out[2].sum()) == 0 and float(out[1].abs().sum()) == 0


def test_sdf_loss_forward_rejects_bad_inputs(dev):
    from omg_planner_amd import _lib, ops
    fx = H.load("cost_topk300.npz")
    poses, sdf, lim, eps, pad, clr, dis = _padded_inputs(fx)
    good = [_t(a, dev) for a in (poses, sdf, lim, np.zeros((4, 3), np.float32), eps, pad, clr, dis)]
    with pytest.raises(_lib.OmgHipError):  # CHECK_CUDA
        ops.sdf_loss_forward(*good[:3], torch.zeros((4, 3)), *good[4:])
    with pytest.raises(_lib.OmgHipError):  # CHECK_CONTIGUOUS
        ops.sdf_loss_forward(*good[:3], torch.zeros((3, 4), device=dev).t(), *good[4:])
    with pytest.raises(_lib.OmgHipError):  # float64 is UB in the reference; an error here
        ops.sdf_loss_forward(*good[:3], torch.zeros((4, 3), dtype=torch.float64, device=dev), *good[4:])
    # raw C ABI: null pointers -> OMGX_ERR_INVALID, never a crash/exit
    rc = _lib.lib().omgx_sdf_loss_forward(None, None, None, None, None, None, None, None, 10, 1, None, None, None, None)
    assert rc == _lib.OMGX_ERR_INVALID


def test_sdf_objects_sum_equals_single_object_runs_full_size(dev):
    """Size-independent property at the C2 size (64 goals x 30 waypoints x 150 points = 288 000 points,
    5 objects): the fused multi-object result equals the in-order float32 sum of single-object runs."""
    from omg_planner_amd import ops, scenes as sc
    scene = sc.make_tabletop_scene(0)
    sdf, lim = sc.pack_padded(scene.objects)
    poses, eps, pad, clr, dis = sc.layer_params(scene, **H.LAYER_CFG)
    rng = np.random.RandomState(2)
    pts = _t(rng.uniform([0.0, -0.6, 0.0], [1.0, 0.6, 0.9], size=(288_000, 3)).astype(np.float32), dev)
    a = [_t(x, dev) for x in (poses, sdf, lim)]
    b = [_t(x, dev) for x in (eps, pad, clr)]
    full = ops.sdf_loss_forward(*a, pts, *b, _t(dis, dev))
    acc = [torch.zeros_like(t) for t in full]
    for o in range(len(scene.objects)):
        only = np.ones_like(dis)
        only[o] = 0
        one = ops.sdf_loss_forward(*a, pts, *b, _t(only, dev))
        acc = [x + y for x, y in zip(acc, one)]
    for f, s in zip(full, acc):
        assert torch.equal(f, s)
    again = ops.sdf_loss_forward(*a, pts, *b, _t(dis, dev))
    assert all(torch.equal(x, y) for x, y in zip(full, again))  # deterministic (no atomics)
    assert float((full[0] > 0).float().mean()) > 0.2


# ------------------------------------------------------------------------------------------------
# (2) omgx_fk_sdf
# ------------------------------------------------------------------------------------------------
def _multi_scene_batch(num_scenes, grid=32):
    from omg_planner_amd import scenes as sc
    scenes = [sc.make_tabletop_scene(s, grid=grid, table_grid=(48, 32, 16)) for s in range(num_scenes)]
    if num_scenes > 2:
        scenes[1].objects[2].name = "floor"       # a disabled object
        scenes[2].objects[0].attached = True      # table override (cost.py:325-328)
    return scenes, sc.pack_table(scenes, H.LAYER_CFG)


@pytest.mark.parametrize("S,Cn,soft", [(1, 30, False), (5, 70, True), (9, 1, False)])
def test_fk_sdf_matches_oracle(dev, S, Cn, soft):
    from omg_planner_amd import ops, robot as rb
    from oracle import oracle as orc
    m = rb.PandaModel(seed=3)
    _, batch = _multi_scene_batch(S)
    rng = np.random.RandomState(S * 100 + Cn)
    lo, hi = m.joint_lower_limit[0], m.joint_upper_limit[0]
    joints = rng.uniform(lo, hi, size=(S, Cn, 9))
    ref = orc.fk_sdf(m.blob(), m.points_per_link, batch, joints, soften_fingers=soft)
    ds = ops.DeviceScenes(batch, dev)
    got = ops.fk_sdf(ops.robot_blob(m, dev), m.points_per_link, ds, _t(joints, dev), soften_fingers=soft)
    torch.cuda.synchronize()
    for name, r, g in zip(("potentials", "grads", "collides"), ref, got):
        g = g.cpu().numpy()
        # identical float32 op; inputs differ only where the float64 FK's last ulp flips a float32 point
        assert (g == r).mean() > 0.999, name
        np.testing.assert_allclose(g, r, rtol=0, atol=5e-6 if name != "grads" else 5e-4, err_msg=name)
    assert ref[0].max() > 0


@pytest.mark.parametrize("case", ["cost_topk1000.npz", "cost_topk300.npz", "cost_finger_n50.npz", "cost_soft_finger.npz"])
def test_fk_sdf_matches_reference_fixture(dev, case):
    from omg_planner_amd import ops
    fx = H.load(case)
    m = H.model_from(fx)
    ds = ops.DeviceScenes(H.batch_from(fx), dev)
    pot, grad, col = ops.fk_sdf(ops.robot_blob(m, dev), m.points_per_link, ds, _t(fx["xi"][None], dev),
                                soften_fingers=int(fx["cfg_uncheck"]) == -1)
    np.testing.assert_allclose(pot[0].cpu().numpy(), fx["potentials"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(grad[0].cpu().numpy(), fx["potential_grads"], rtol=0, atol=2e-4)
    assert float(col.sum()) == float(fx["collide_sum"])


# ------------------------------------------------------------------------------------------------
# (3) omgx_goalset_cost
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["arc_g6_n30", "arc_g5_n7", "arc_attached_g4_n12"])
def test_goalset_cost_matches_reference_fixture(dev, case):
    from omg_planner_amd import ops
    fx = H.load(f"batch_{case}.npz")
    m = H.model_from(fx)
    n, G = int(fx["n_remaining"]), fx["goals"].shape[0]
    ds = ops.DeviceScenes(H.batch_from(fx), dev)
    cost, col, pots = ops.goalset_cost(ops.robot_blob(m, dev), m.points_per_link, ds, _t(fx["traj_start"][None], dev),
                                       _t(fx["goals"][None], dev), n, float(fx["cfg_dt"]), soften_fingers=int(fx["uncheck"]) == -1,
                                       want_potentials=True)
    ref = fx["potentials"].reshape(G, n, 10, -1)
    np.testing.assert_allclose(pots[0].cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(cost[0].cpu().numpy(), fx["goal_cost"], rtol=1e-5, atol=1e-6)  # north_star: 1e-4
    assert float(col.sum()) == float(fx["collides"].sum())


@pytest.mark.parametrize("S,G,n", [(3, 7, 30), (10, 4, 1), (2, 33, 13)])
def test_goalset_cost_matches_oracle(dev, S, G, n):
    from omg_planner_amd import ops, robot as rb, scenes as sc
    from oracle import oracle as orc
    m = rb.PandaModel(seed=4)
    _, batch = _multi_scene_batch(S)
    starts = np.stack([sc.cubic_init(rb.HOME_CONFIG, sc.make_goal_set(s, 1)[0], 30)[30 - n] for s in range(S)])
    goals = np.stack([sc.make_goal_set(s, G) for s in range(S)])
    ref_cost, ref_col, ref_pots = orc.goalset_cost(m.blob(), m.points_per_link, batch, starts, goals, n, 0.1, want_potentials=True)
    ds = ops.DeviceScenes(batch, dev)
    cost, col, pots = ops.goalset_cost(ops.robot_blob(m, dev), m.points_per_link, ds, _t(starts, dev), _t(goals, dev), n, 0.1,
                                       want_potentials=True)
    g = pots.cpu().numpy()
    assert (g == ref_pots).mean() > 0.999
    np.testing.assert_allclose(g, ref_pots, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(cost.cpu().numpy(), ref_cost, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(col.cpu().numpy(), ref_col)


# ------------------------------------------------------------------------------------------------
# (4) omgx_chomp_optimize
# ------------------------------------------------------------------------------------------------
COST_CASES = ["topk1000", "topk300", "clean", "fixed_end", "soft_finger", "finger_n50", "attached", "short_n5"]
OPT_CASES = ["standoff_20", "nostandoff_20", "fixed_end_5", "limits_5", "n50_dt006_5"]


@pytest.mark.parametrize("case", COST_CASES)
def test_total_loss_matches_reference_fixture(dev, case):
    """Cost.compute_total_loss: cost, gradient and info of the reference (info_only step)."""
    from omg_planner_amd import _lib, ops
    fx = H.load(f"cost_{case}.npz")
    m = H.model_from(fx)
    n, P = fx["xi"].shape[0], m.points_per_link
    robot = ops.robot_blob(m, dev)
    ds = ops.DeviceScenes(H.batch_from(fx), dev)
    traj = _t(fx["xi"][None], dev)
    _, _, col = ops.fk_sdf(robot, P, ds, traj, soften_fingers=int(fx["cfg_uncheck"]) == -1)
    prm = H.params_from(fx, _lib.ChompParams, n, P, 0, float(fx["cfg_obstacle_weight"]), float(fx["cfg_smoothness_weight"]))
    goal = _t(np.tile(fx["end"], (1, prm.constraint_num, 1)), dev)
    # the reference's own layer outputs as inputs (isolates the float64 stage)
    grad, cost_traj, info = ops.chomp_optimize(robot, prm, traj, _t(fx["start"][None], dev), _t(fx["end"][None], dev), goal,
                                               _t(fx["goal_point"][None], dev), _t(fx["potentials"][None], dev),
                                               _t(fx["potential_grads"][None], dev), col)
    np.testing.assert_array_equal(traj[0].cpu().numpy(), fx["xi"])  # info_only leaves the trajectory alone
    np.testing.assert_allclose(grad[0].cpu().numpy(), fx["total_grad"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(cost_traj[0].cpu().numpy(), fx["info_cost_traj"], rtol=1e-9, atol=1e-9)
    info = info[0].cpu().numpy()
    for key in ["cost", "obs", "smooth", "weighted_obs", "weighted_smooth", "weighted_obs_grad", "weighted_smooth_grad",
                "grad", "collide", "reach", "standoff_idx", "terminate", "failure_terminate", "execute"]:
        np.testing.assert_allclose(info[H.INFO_IDX[key]], fx["info_" + key], rtol=1e-9, atol=1e-9, err_msg=key)


@pytest.mark.parametrize("case", OPT_CASES)
def test_optimizer_sequence_matches_reference_fixture(dev, case):
    """Optimizer.optimize for k consecutive steps: teacher-forced 1e-9, free-running 1e-6 (bar: 1e-4)."""
    from omg_planner_amd import _lib, ops
    fx = H.load(f"opt_{case}.npz")
    m = H.model_from(fx)
    hist = fx["traj_history"]
    steps, n, P = hist.shape[0] - 1, hist.shape[1], m.points_per_link
    robot = ops.robot_blob(m, dev)
    ds = ops.DeviceScenes(H.batch_from(fx), dev)
    gi = int(fx["goal_idx"])
    goal = _t((fx["reach_grasps"][gi] if int(fx["cfg_use_standoff"]) else fx["goal_set"][gi][None])[None], dev)
    goal_point = _t(fx["goal_set"][gi][None], dev)
    start, end = _t(fx["start"][None], dev), _t(fx["end"][None], dev)
    free = _t(hist[0][None], dev)
    for k in range(steps + 1):
        w_obs, w_sm, eta = fx["schedule"][k]
        upd = (1 if int(fx.get("cfg_force_update", 1)) else 2) if k < steps else 0  # force_update=False -> do_update 2
        prm = H.params_from(fx, _lib.ChompParams, n, P, upd, w_obs, w_sm, eta, int(fx["cfg_reach_tail_length"]))
        for mode in ("forced", "free"):
            traj = _t(hist[k][None], dev) if mode == "forced" else free
            pot, pg, col = ops.fk_sdf(robot, P, ds, traj)
            grad, _, info = ops.chomp_optimize(robot, prm, traj, start, end, goal, goal_point, pot, pg, col)
            if mode == "forced":
                np.testing.assert_allclose(grad[0].cpu().numpy(), fx["info_gradient"][k], rtol=1e-7, atol=1e-7, err_msg=f"step {k}")
                inf = info[0].cpu().numpy()
                for key in ["cost", "obs", "smooth", "collide", "reach", "terminate", "violate_limit", "execute", "failure_terminate"]:
                    np.testing.assert_allclose(inf[H.INFO_IDX[key]], fx["info_" + key][k], rtol=1e-7, atol=1e-7, err_msg=f"{key} step {k}")
            if upd:
                np.testing.assert_allclose(traj[0].cpu().numpy(), hist[k + 1], rtol=0, atol=1e-9 if mode == "forced" else 1e-6,
                                           err_msg=f"{mode} step {k}")


def test_chomp_optimize_batched_matches_oracle(dev):
    """S scenes at once (mixed scenes, inactive mask) against the oracle, 3 consecutive steps."""
    from omg_planner_amd import _lib, ops, robot as rb, scenes as sc
    from oracle import oracle as orc
    S, n = 12, 30
    m = rb.PandaModel(seed=5)
    P = m.points_per_link
    _, batch = _multi_scene_batch(S)
    goals = np.stack([sc.make_goal_set(s, 1)[0] for s in range(S)])
    traj0 = np.stack([sc.cubic_init(rb.HOME_CONFIG, goals[s], n) for s in range(S)])
    start = np.tile(rb.HOME_CONFIG, (S, 1))
    active = np.ones(S, np.int32)
    active[3] = 0
    fx = dict(cfg_top_k=1000, cfg_goal_set_proj=1, cfg_use_standoff=0, cfg_dt=0.1)
    robot, ds = ops.robot_blob(m, dev), ops.DeviceScenes(batch, dev)
    t_dev = _t(traj0, dev)
    t_ref = traj0.copy()
    for step in range(1, 4):
        w_sm = 0.1 * 1.02 ** step
        po = H.params_from(fx, orc.ChompParams, n, P, 1, 1.0, w_sm)
        pd = H.params_from(fx, _lib.ChompParams, n, P, 1, 1.0, w_sm)
        rp, rg, rc = orc.fk_sdf(m.blob(), P, batch, t_ref)
        t_ref, g_ref, ct_ref, info_ref = orc.chomp_optimize(m.blob(), po, t_ref, start, goals, goals[:, None], goals, rp, rg, rc, active)
        pot, pg, col = ops.fk_sdf(robot, P, ds, t_dev)
        g, ct, info = ops.chomp_optimize(robot, pd, t_dev, _t(start, dev), _t(goals, dev), _t(goals[:, None], dev), _t(goals, dev),
                                         pot, pg, col, active=_t(active, dev))
        act = active.astype(bool)
        np.testing.assert_allclose(t_dev.cpu().numpy(), t_ref, rtol=0, atol=1e-7)
        np.testing.assert_allclose(g.cpu().numpy()[act], g_ref[act], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(info.cpu().numpy()[act], info_ref[act], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(t_dev[3].cpu().numpy(), traj0[3])  # inactive trajectory untouched


# ------------------------------------------------------------------------------------------------
# (5) the reference's class surface: Cost / Optimizer / omg_cuda
# ------------------------------------------------------------------------------------------------
class _Traj:
    """The attributes of omg.core.Trajectory the path reads (core.py:23-57)."""

    def __init__(self, data, start, end, goal_set, goal_idx=0):
        self.data, self.start, self.end = np.array(data), np.array(start), np.array(end)
        self.goal_set, self.goal_idx = goal_set, goal_idx

    def set(self, new):
        self.data = new


def _env_from(fx, dev, cfg):
    import types
    m = H.model_from(fx)
    robot = types.SimpleNamespace(collision_points=m.collision_points, joint_lower_limit=m.joint_lower_limit,
                                  joint_upper_limit=m.joint_upper_limit)
    objs = [types.SimpleNamespace(name=str(n), pose_mat=fx["obj_pose"][i], attached=bool(fx["attached"][i]), reach_grasps=[])
            for i, n in enumerate(fx["obj_names"])]
    return types.SimpleNamespace(robot=robot, objects=objs, target_idx=int(fx["target_idx"]), config=cfg,
                                 sdf_torch=_t(fx["sdf"], dev), sdf_limits=_t(fx["limits"], dev))


def _cfg_from(fx, n):
    from omg_planner_amd.config import Config
    cfg = Config(timesteps=n, top_k_collision=int(fx["cfg_top_k"]), goal_set_proj=bool(fx["cfg_goal_set_proj"]),
                 use_standoff=bool(fx["cfg_use_standoff"]), consider_finger=bool(fx.get("cfg_consider_finger", 0)),
                 uncheck_finger_collision=int(fx.get("cfg_uncheck", 0)))
    cfg.time_interval = float(fx["cfg_dt"])
    cfg._mats = None
    for key in ("allow_collision_point", "pre_terminate", "terminate_smooth_loss", "clip_grad_scale", "joint_limit_max_steps"):
        if "cfg_" + key in fx:  # fixtures from tests/fuzz/make_random_fixtures.py vary them
            setattr(cfg, key, type(getattr(cfg, key))(fx["cfg_" + key]))
    return cfg


@pytest.mark.parametrize("case", ["topk1000", "clean", "fixed_end", "finger_n50"])
def test_cost_class_matches_reference_fixture(dev, case):
    from omg_planner_amd.cost import Cost
    fx = H.load(f"cost_{case}.npz")
    n = fx["xi"].shape[0]
    cfg = _cfg_from(fx, n)
    cfg.obstacle_weight, cfg.smoothness_weight = float(fx["cfg_obstacle_weight"]), float(fx["cfg_smoothness_weight"])
    cost = Cost(_env_from(fx, dev, cfg))
    traj = _Traj(fx["xi"], fx["start"], fx["end"], fx["goal_point"][None])
    total, grad, info = cost.compute_total_loss(traj)
    np.testing.assert_allclose(total, fx["total_cost"], rtol=1e-6)        # north_star bar: 1e-4
    np.testing.assert_allclose(grad, fx["total_grad"], rtol=1e-5, atol=1e-5)
    for key in ["obs", "smooth", "weighted_obs", "weighted_smooth", "collide", "reach", "standoff_idx", "terminate", "execute"]:
        np.testing.assert_allclose(float(info[key]), float(fx["info_" + key]), rtol=1e-6, atol=1e-6, err_msg=key)
    obs_cost, obs_grad, _, col = cost.compute_collision_loss(fx["xi"], fx["start"], fx["end"])
    np.testing.assert_allclose(obs_cost, fx["obs_cost"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(obs_grad, fx["obs_grad"], rtol=1e-5, atol=1e-5)
    sm_loss, sm_grad = cost.compute_smooth_loss(fx["xi"], fx["start"], fx["end"])
    np.testing.assert_allclose(sm_loss, fx["smooth_loss"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(sm_grad, fx["smooth_grad"], rtol=1e-9, atol=1e-7)


def test_cost_forward_poses_matches_reference_fk(dev):
    from omg_planner_amd.config import Config
    from omg_planner_amd.cost import Cost
    fk = H.load("fk.npz")
    fx = H.load("cost_topk1000.npz")
    cost = Cost(_env_from(fx, dev, Config()))
    for b in range(4):
        q = fk["joints"][b]
        deg = np.rad2deg(np.concatenate([q[:7], [0.0], q[7:]]))  # wrap_value
        poses, org, ax = cost.forward_poses(deg)
        np.testing.assert_allclose(poses, fk["poses"][b], atol=1e-12)
        np.testing.assert_allclose(org, fk["joint_origins"][b], atol=1e-12)
        np.testing.assert_allclose(ax, fk["joint_axis"][b], atol=1e-12)


@pytest.mark.parametrize("case", ["arc_g6_n30", "arc_g5_n7", "noarc_soft_g8"])
def test_cost_batch_obstacle_cost_matches_reference_fixture(dev, case):
    from omg_planner_amd.config import Config
    from omg_planner_amd.cost import Cost
    fx = H.load(f"batch_{case}.npz")
    cfg = Config(timesteps=30)
    cost = Cost(_env_from(fx, dev, cfg))
    arc = int(fx["n_remaining"]) if int(fx["arc_length"]) else -1
    pot, grad, vis, col = cost.batch_obstacle_cost(fx["joints"], arc_length=arc, special_check_id=0,
                                                   uncheck_finger_collision=int(fx["uncheck"]), start=fx["traj_start"], end=fx["goals"])
    np.testing.assert_allclose(pot.cpu().numpy(), fx["potentials"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(grad.cpu().numpy(), fx["grads"], rtol=0, atol=2e-4)
    np.testing.assert_array_equal(col.cpu().numpy(), fx["collides"])
    assert vis.shape == (fx["joints"].shape[0], 10, 15, 12)


@pytest.mark.parametrize("case", ["standoff_20", "fixed_end_5", "limits_5"])
def test_optimizer_class_matches_reference_fixture(dev, case):
    """Optimizer.optimize through the class surface, free-running from the reference's start: 1e-6 (bar 1e-4)."""
    from omg_planner_amd.cost import Cost
    from omg_planner_amd.optimizer import Optimizer
    import types
    fx = H.load(f"opt_{case}.npz")
    hist = fx["traj_history"]
    steps, n = hist.shape[0] - 1, hist.shape[1]
    fx2 = dict(fx)
    fx2.setdefault("cfg_consider_finger", 0)
    cfg = _cfg_from(fx2, n)
    env = _env_from(fx, dev, cfg)
    env.objects[env.target_idx].reach_grasps = fx["reach_grasps"]
    cost = Cost(env)
    opt = Optimizer(types.SimpleNamespace(config=cfg, robot=env.robot), cost)
    traj = _Traj(hist[0], fx["start"], fx["end"], fx["goal_set"], int(fx["goal_idx"]))
    for k in range(steps):
        info = opt.optimize(traj, force_update=bool(int(fx.get("cfg_force_update", 1))))
        np.testing.assert_allclose(traj.data, hist[k + 1], rtol=0, atol=1e-6, err_msg=f"step {k}")
        np.testing.assert_allclose(info["cost"], fx["info_cost"][k], rtol=1e-5, err_msg=f"step {k}")
        assert bool(info["violate_limit"]) == bool(fx["info_violate_limit"][k])
    final = opt.optimize(traj, info_only=True)
    np.testing.assert_allclose(final["cost"], fx["info_cost"][steps], rtol=1e-5)
    np.testing.assert_allclose(np.array(fx["schedule"][steps]), [cfg.obstacle_weight, cfg.smoothness_weight, cfg.step_size])


def test_cost_object_table_follows_the_environment(dev):
    """Cost caches its device object table between calls; every input the reference re-reads per call (cost.py:303-328)
    must invalidate it: a moved object, a changed cfg.epsilon, another target, an in-place edit of env.sdf_torch /
    env.sdf_limits, a replaced tensor.  Each time the result must equal that of a fresh Cost on the same environment."""
    from omg_planner_amd.cost import Cost
    fx = H.load("cost_topk300.npz")
    n = fx["xi"].shape[0]
    cfg = _cfg_from(fx, n)
    cfg.obstacle_weight, cfg.smoothness_weight = float(fx["cfg_obstacle_weight"]), float(fx["cfg_smoothness_weight"])
    env = _env_from(fx, dev, cfg)
    cost = Cost(env)
    traj = _Traj(fx["xi"], fx["start"], fx["end"], fx["goal_point"][None])

    def both():
        a = cost.compute_total_loss(traj)
        b = Cost(env).compute_total_loss(traj)
        assert a[0] == b[0] and np.array_equal(a[1], b[1])
        return a[0]

    seen = [both()]
    np.testing.assert_allclose(seen[0], fx["total_cost"], rtol=1e-6)
    assert cost.compute_total_loss(traj)[0] == seen[0]  # cached table, same answer
    env.objects[1].pose_mat = env.objects[1].pose_mat.copy()
    env.objects[1].pose_mat[:3, 3] += [0.05, -0.03, 0.02]  # an object moves
    seen.append(both())
    env.objects[2].pose_mat[:3, 3] -= 0.04                 # in place
    seen.append(both())
    cfg.epsilon = 0.3
    seen.append(both())
    env.target_idx = (env.target_idx + 1) % len(env.objects)
    seen.append(both())
    env.sdf_torch[1] += 0.02                               # in-place edit of the volumes (version counter)
    seen.append(both())
    env.sdf_torch = env.sdf_torch.clone() - 0.01           # a new tensor
    seen.append(both())
    env.objects[0].name = "floor"                          # disabled
    seen.append(both())
    # a write through a RAW device pointer (the C ABI, e.g. omgx_point_cloud_sdf into a slice of env.sdf_torch) does not move
    # the version counter: the cached influence boxes of the old volume would cull the new one.  ops.point_cloud_sdf(out=...)
    # bumps the counter itself; after any other raw write Cost.invalidate() does it.
    from omg_planner_amd import ops
    vol = env.sdf_torch[1]
    cloud = torch.as_tensor(np.random.RandomState(2).uniform(0.2, 0.3, (50, 3)), dtype=torch.float64, device=dev)
    lo = (cloud.min(0).values.cpu().numpy() - 0.24)
    hi = (cloud.max(0).values.cpu().numpy() + 0.24)
    dims = [len(np.arange(lo[a], hi[a], 0.02)) for a in range(3)]
    if int(np.prod(dims)) <= vol.numel():
        flat = vol.reshape(-1)[: int(np.prod(dims))]
        v0 = env.sdf_torch._version
        ops.point_cloud_sdf(cloud, out=flat)
        assert env.sdf_torch._version > v0
        seen.append(both())
    ctypes_write = env.sdf_torch.data_ptr()  # what a C caller holds: emulate its write without touching the counter
    with torch.no_grad():
        raw = torch.empty(0, dtype=torch.float32, device=dev).set_(env.sdf_torch.untyped_storage(), 0, (env.sdf_torch.numel(),))
        v1 = env.sdf_torch._version
        raw.detach().mul_(0.5)
    assert ctypes_write == env.sdf_torch.data_ptr()
    cost.invalidate()
    seen.append(both())
    assert len(set(seen)) == len(seen), seen


def test_visualisation_arrays_match_reference(dev):
    """info["collision_pts"] of compute_total_loss / Optimizer.optimize (built on first access) and vis_pts of
    batch_obstacle_cost against the reference's arrays (tests/golden/vis.npz): shape [n,10,p,12], positions, colours from
    color_point on the un-weighted potentials + the top-k highlight, gradients."""
    from omg_planner_amd.config import Config
    from omg_planner_amd.cost import Cost
    from omg_planner_amd.optimizer import Optimizer
    import types
    fx = H.load("vis.npz")

    def sub(prefix):
        d = {k[len(prefix):]: fx[k] for k in fx if k.startswith(prefix)}
        d["collision_points"] = fx["collision_points"]
        return d

    def compare(got, ref, what):
        assert got.shape == ref.shape, what
        np.testing.assert_allclose(got[..., :3], ref[..., :3], rtol=0, atol=1e-6, err_msg=what + " positions")
        np.testing.assert_allclose(got[..., 9:], ref[..., 9:], rtol=0, atol=2e-4, err_msg=what + " gradients")
        np.testing.assert_array_equal(got[..., 3:6], ref[..., 3:6])
        # colours: 255 * relative potential; a float32 point that flips by one ulp moves a potential by ~1e-6
        bad = np.abs(got[..., 6:9] - ref[..., 6:9]).max(-1) > 0.05
        assert bad.mean() < 2e-3, f"{what}: {int(bad.sum())} of {bad.size} colours differ"

    for tag in ("topk", "clean", "soft"):
        d = sub(tag + "_")
        cfg = Config(timesteps=30, top_k_collision=int(d["top_k"]), uncheck_finger_collision=int(d["uncheck"]))
        cfg.obstacle_weight, cfg.smoothness_weight = cfg.base_obstacle_weight, cfg.smoothness_base_weight * cfg.cost_schedule_boost
        env = _env_from(d, dev, cfg)
        cost = Cost(env)
        traj = _Traj(d["xi"], fx["start"], fx["end"], d["goal_point"][None])
        _, _, info = cost.compute_total_loss(traj)
        assert "collision_pts" not in info.keys() and len(info) == 19  # lazy: built by the first info["collision_pts"]
        compare(info["collision_pts"], d["collision_pts"], tag)
        assert "collision_pts" in info and len(info) == 20
        cfg.use_standoff = False  # no reach_grasps in this fixture; the layer outputs do not depend on it
        opt = Optimizer(types.SimpleNamespace(config=cfg, robot=env.robot), cost)
        info2 = opt.optimize(traj, info_only=True)  # Optimizer.update moves the weights, not the layer outputs
        compare(info2["collision_pts"], d["collision_pts"], tag + " (optimize)")
        _, _, vis, _ = cost.compute_collision_loss(d["xi"], fx["start"], fx["end"])
        compare(vis, d["collision_pts"], tag + " (compute_collision_loss)")
    d = sub("batch_")
    cost = Cost(_env_from(d, dev, Config(timesteps=30)))
    _, _, vis, _ = cost.batch_obstacle_cost(d["joints"], arc_length=7, special_check_id=0, uncheck_finger_collision=0,
                                            start=d["traj_start"], end=d["goals"])
    compare(vis, d["vis_pts"], "batch")


def test_omg_cuda_module_is_a_drop_in(dev):
    """`import omg_cuda; omg_cuda.sdf_loss_forward(...)` as layers/sdf_matching_loss.py:21-30 calls it."""
    import importlib
    import sys
    from pathlib import Path
    from oracle import oracle as orc
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "omg-planner_amd"))
    try:
        omg_cuda = importlib.import_module("omg_cuda")
    finally:
        sys.path.pop(0)
    fx = H.load("cost_topk300.npz")
    poses, sdf, lim, eps, pad, clr, dis = _padded_inputs(fx)
    pts = np.random.RandomState(5).uniform([-0.2, -0.6, 0.0], [1.0, 0.6, 1.0], size=(5000, 3)).astype(np.float32)
    out = omg_cuda.sdf_loss_forward(*[_t(a, dev) for a in (poses, sdf, lim, pts, eps, pad, clr, dis)])
    ref = orc.sdf_loss_forward(poses, sdf, lim, pts, eps, pad, clr, dis)
    assert len(out) == 3
    for r, g in zip(ref, out):
        np.testing.assert_array_equal(g.cpu().numpy(), r)


# ------------------------------------------------------------------------------------------------
# (6) omgx_goal_update — Learner.update_goal
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["FTL_0", "FTC_0", "Exp_0", "MD_0", "MD_1", "FTC_0_close", "MD_0_close"])
def test_goal_update_matches_reference_learner_fixture(dev, case):
    from omg_planner_amd import _lib, ops
    from tests.test_oracle_learner import learner_params
    fx = H.load(f"learner_{case}.npz")
    m = H.model_from(fx)
    G = fx["goal_set"].shape[0]
    standoff = int(fx["cfg_use_standoff"])
    c = fx["reach_grasps"].shape[1] if standoff else 1
    robot, ds = ops.robot_blob(m, dev), ops.DeviceScenes(H.batch_from(fx), dev)
    goal_set, reach = _t(fx["goal_set"][None], dev), _t(fx["reach_grasps"][None], dev)
    cv_goals = _t((fx["reach_grasps"][:, -1, :] if standoff else fx["goal_set"])[None], dev)
    state = ops.learner_state(1, G, dev)
    idx = torch.zeros(1, dtype=torch.int32, device=dev)
    end, rows, gp = (torch.zeros((1, 9), dtype=torch.float64, device=dev), torch.zeros((1, c, 9), dtype=torch.float64, device=dev),
                     torch.zeros((1, 9), dtype=torch.float64, device=dev))
    cv = torch.zeros((1, G), dtype=torch.float64, device=dev)
    for k in range(fx["trajs"].shape[0]):
        po = learner_params(fx, k + 1)
        prm = _lib.LearnerParams()
        for f, _ in prm._fields_:
            setattr(prm, f, getattr(po, f))
        traj = _t(fx["trajs"][k][None], dev)
        cost, _, _ = ops.goalset_cost(robot, m.points_per_link, ds, traj[:, prm.start_idx].contiguous(), cv_goals,
                                      prm.n_waypoints - prm.start_idx, float(fx["cfg_dt"]))
        ops.goal_update(prm, traj, goal_set, reach, cost, state, idx, end, rows, gp, cv)
        np.testing.assert_allclose(cv[0].cpu().numpy(), fx["cost_vectors"][k], rtol=2e-5, atol=1e-7, err_msg=f"cv step {k}")
        np.testing.assert_allclose(state[0, G:2 * G].cpu().numpy(), fx["p"][k], rtol=1e-4, atol=1e-6, err_msg=f"p step {k}")
        assert int(idx[0]) == int(fx["goal_idx"][k])
        np.testing.assert_array_equal(end[0].cpu().numpy(), fx["goal_set"][int(idx[0])])
        exp_rows = fx["reach_grasps"][int(idx[0])] if standoff else fx["goal_set"][int(idx[0])][None]
        np.testing.assert_array_equal(rows[0].cpu().numpy(), exp_rows)
        if str(fx["alg"]) == "MD":
            np.testing.assert_allclose(state[0, 7 * G:7 * G + 5].cpu().numpy(), fx["q"][k], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("case", ["FTL_0", "FTC_0", "Exp_0", "MD_0", "MD_1", "FTC_0_close", "MD_0_close", "MD_0_reset", "Exp_0_reset"])
def test_learner_class_matches_reference_learner_fixture(dev, case):
    """The mirror class online_learner.Learner driven like the reference's (constructor picks the initial goal; then
    update_goal per step on the trajectories the fixture recorded): goal indices and the public attributes p, q, sum_costs,
    experts_p, cost_vector against the reference Learner's own (tests/golden/learner_*.npz)."""
    from omg_planner_amd.config import Config
    from omg_planner_amd.cost import Cost
    from omg_planner_amd.online_learner import Learner
    fx = H.load(f"learner_{case}.npz")
    alg, standoff = str(fx["alg"]), bool(int(fx["cfg_use_standoff"]))
    cfg = Config(timesteps=30, use_standoff=standoff, ol_alg=alg, optim_steps=int(fx["optim_steps"]), dist_eps=float(fx["dist_eps"]),
                 normalize_cost=bool(int(fx.get("cfg_normalize_cost", 1))), base_obstacle_weight=float(fx.get("cfg_base_obstacle_weight", 1.0)),
                 smoothness_base_weight=float(fx.get("cfg_smoothness_base_weight", 0.1)))
    env = _env_from(fx, dev, cfg)
    env.objects[env.target_idx].reach_grasps = fx["reach_grasps"]
    traj = _Traj(fx["traj"], fx["start"], fx["goal_set"][0], fx["goal_set"], 0)
    traj.interpolate_waypoints = lambda *a, **k: None  # the generator kept the trajectory fixed across Learner.__init__ too
    learner = Learner(env, traj, Cost(env))
    assert int(traj.goal_idx) == int(fx["init_goal_idx"]) and abs(learner.eta - float(fx["eta"])) < 1e-15
    np.testing.assert_array_equal(traj.end, fx["goal_set"][int(fx["init_goal_idx"])])
    reset_at = int(fx["reset_at"]) if "reset_at" in fx else -1
    for k in range(fx["trajs"].shape[0]):
        if k == reset_at:  # Learner.reset(traj) for a new trajectory object over the same goal set (omg/online_learner.py:251-263)
            gi = int(traj.goal_idx)
            traj = _Traj(fx["trajs"][k], fx["start"], fx["goal_set"][gi], fx["goal_set"], gi)
            traj.interpolate_waypoints = lambda *a, **k: None
            learner.reset(traj)
            assert learner.t == 0.0 and learner.traj is traj and learner.last_leader == 0
            np.testing.assert_array_equal(learner.p, np.ones(len(fx["goal_set"])) / len(fx["goal_set"]))
        traj.data = fx["trajs"][k]
        learner.t += 1
        cv = learner.cost_vector()
        learner.t -= 1
        np.testing.assert_allclose(cv, fx["cost_vectors"][k], rtol=2e-5, atol=1e-7, err_msg=f"cost vector step {k}")
        learner.update_goal()
        assert int(traj.goal_idx) == int(fx["goal_idx"][k]), k
        np.testing.assert_array_equal(traj.end, fx["goal_set"][int(fx["goal_idx"][k])])
        np.testing.assert_allclose(learner.p, fx["p"][k], rtol=1e-4, atol=1e-6, err_msg=f"p step {k}")
        if alg == "MD":
            np.testing.assert_allclose(learner.q, fx["q"][k], rtol=1e-4, atol=1e-7, err_msg=f"q step {k}")
    assert learner.t == float(fx["final_t"])
    if alg in ("FTL", "Exp"):
        np.testing.assert_allclose(learner.sum_costs, fx["sum_costs"], rtol=2e-5)
    if alg == "MD":
        np.testing.assert_allclose(np.stack(learner.experts_p), fx["experts_p"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("case", ["FTL_0", "MD_1", "FTC_0_close"])
def test_engine_initial_goal_matches_reference_learner_init(dev, case):
    """ChompEngine.select_initial_goal = Learner.__init__ (online_learner.py:96-102): argmin of the t = 0 cost vector on the
    initial trajectory, traj.end <- that goal, clamped-cubic re-interpolation (Trajectory.interpolate_waypoints)."""
    from omg_planner_amd import scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    fx = H.load(f"learner_{case}.npz")
    standoff = bool(int(fx["cfg_use_standoff"]))
    cfg = Config(timesteps=30, use_standoff=standoff)
    eng = ChompEngine(H.model_from(fx), H.batch_from(fx), cfg, fx["start"][None], fx["goal_set"][None],
                      reach_grasps=fx["reach_grasps"][None] if standoff else None, traj_init=fx["traj"][None], device=dev,
                      ol_alg=str(fx["alg"]))
    state0 = eng.learner_state.clone()
    eng.select_initial_goal()
    gi = int(fx["init_goal_idx"])
    assert int(eng.goal_idx[0]) == gi
    np.testing.assert_array_equal(eng.end[0].cpu().numpy(), fx["goal_set"][gi])
    np.testing.assert_allclose(eng.traj[0].cpu().numpy(), sc.cubic_init(fx["start"], fx["goal_set"][gi], 30), rtol=0, atol=1e-12)
    assert torch.equal(eng.learner_state, state0) and eng.t == 0  # the pick leaves the learner untouched


def test_engine_proj_and_baseline_keep_a_fixed_goal(dev):
    """planner.py:200-222, 609-618: with ol_alg "Proj" the goal is the one closest to the START (link_smooth_weight metric),
    with "Baseline" cfg.goal_idx; neither runs the learner during the plan."""
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    S, G, n = 3, 7, 12
    m = rb.PandaModel(seed=11)
    scenes, batch = _multi_scene_batch(S)
    goals = np.stack([sc.make_goal_set(s, G) for s in range(S)])
    start = np.tile(rb.HOME_CONFIG, (S, 1)) + np.random.RandomState(0).normal(0, 0.1, (S, 9)) * np.array([1] * 7 + [0, 0])
    for alg, expect in (("Proj", np.argmin(np.linalg.norm(start[:, None] - goals, axis=-1), axis=1)), ("Baseline", np.zeros(S, int))):
        cfg = Config(use_standoff=False)
        cfg.get_global_param(n)
        eng = ChompEngine(m, batch, cfg, start, goals, device=dev, ol_alg=alg)
        eng.select_initial_goal()
        np.testing.assert_array_equal(eng.goal_idx.cpu().numpy(), expect)
        np.testing.assert_array_equal(eng.end.cpu().numpy(), goals[np.arange(S), expect])
        np.testing.assert_allclose(eng.traj.cpu().numpy(), np.stack([sc.cubic_init(start[s], goals[s, expect[s]], n) for s in range(S)]),
                                   rtol=0, atol=1e-12)
        for t in range(3):
            eng.iterate(t)
        np.testing.assert_array_equal(eng.goal_idx.cpu().numpy(), expect)  # no learner in the loop
        assert eng.t == 0


@pytest.mark.parametrize("alg,G", [("MD", 64), ("Exp", 100), ("FTL", 200), ("Proj", 33), ("MD", 130)])
def test_goal_update_matches_oracle_many_scenes(dev, alg, G):
    from omg_planner_amd import _lib, ops
    from oracle import oracle as orc
    S, n, c = 7, 30, 5
    rng = np.random.RandomState(G)
    traj = rng.normal(0, 0.5, size=(S, n, 9))
    goal_set = rng.normal(0, 0.5, size=(S, G, 9))
    reach = rng.normal(0, 0.5, size=(S, G, c, 9))
    po = orc.LearnerParams()
    po.alg, po.num_goals, po.n_waypoints, po.start_idx, po.constraint_num, po.use_standoff = orc.ALG[alg], G, n, 4, c, 1
    po.normalize_cost, po.base_obstacle_weight, po.smooth_weight, po.eta = 1, 1.0, 0.01, float(np.sqrt(np.log(G + 1) / 50))
    pd = _lib.LearnerParams()
    for f, _ in pd._fields_:
        setattr(pd, f, getattr(po, f))
    st_ref = orc.learner_state_init(S, G)
    st = ops.learner_state(S, G, dev)
    idx = torch.zeros(S, dtype=torch.int32, device=dev)
    end, rows, gp = (torch.zeros((S, 9), dtype=torch.float64, device=dev), torch.zeros((S, c, 9), dtype=torch.float64, device=dev),
                     torch.zeros((S, 9), dtype=torch.float64, device=dev))
    for step in range(4):
        gc = rng.gamma(2.0, 2.0, size=(S, G)).astype(np.float32)
        r_idx, r_end, r_rows, r_gp, _ = orc.goal_update(po, traj, goal_set, reach, gc, st_ref)
        ops.goal_update(pd, _t(traj, dev), _t(goal_set, dev), _t(reach, dev), _t(gc, dev), st, idx, end, rows, gp)
        np.testing.assert_array_equal(idx.cpu().numpy(), r_idx)
        np.testing.assert_allclose(st.cpu().numpy(), st_ref, rtol=1e-5, atol=1e-8)
        np.testing.assert_array_equal(rows.cpu().numpy(), r_rows)
        np.testing.assert_array_equal(end.cpu().numpy(), r_end)


@pytest.mark.parametrize("alg", ["FTL", "FTC", "Exp", "MD"])
def test_goal_update_degenerate_cost_vector_keeps_a_valid_index(dev, alg):
    """Zero cost vector -> 0/0 = NaN after normalisation: numpy's argmin/argmax pick the first NaN (goal 0); the kernel must
    not leave the index undefined (out-of-bounds goal gather; found by tests/fuzz/fuzz_parity.py)."""
    from omg_planner_amd import _lib, ops
    from oracle import oracle as orc
    S, G, n = 3, 5, 12
    rng = np.random.RandomState(2)
    traj = rng.uniform(-1, 1, (S, n, 9))
    goals = np.repeat(traj[:, 7][:, None, :], G, axis=1)  # every goal == traj_start
    prm = _lib.LearnerParams()
    prm.alg, prm.num_goals, prm.n_waypoints, prm.start_idx = _lib.ALG[alg], G, n, 7
    prm.constraint_num, prm.use_standoff, prm.normalize_cost = 1, 0, 1
    prm.base_obstacle_weight, prm.smooth_weight, prm.eta = 1.0, 0.01, 0.3
    po = orc.LearnerParams()
    for f, _ in po._fields_:
        setattr(po, f, getattr(prm, f))
    st_ref = orc.learner_state_init(S, G)
    r_idx, r_end, r_rows, r_gp, r_cv = orc.goal_update(po, traj, goals, None, np.zeros((S, G), np.float32), st_ref)
    st = ops.learner_state(S, G, dev)
    idx = torch.full((S,), -7, dtype=torch.int32, device=dev)
    end, rows, gp = (torch.zeros((S, 9), dtype=torch.float64, device=dev), torch.zeros((S, 1, 9), dtype=torch.float64, device=dev),
                     torch.zeros((S, 9), dtype=torch.float64, device=dev))
    cv = torch.zeros((S, G), dtype=torch.float64, device=dev)
    ops.goal_update(prm, _t(traj, dev), _t(goals, dev), None, torch.zeros((S, G), dtype=torch.float32, device=dev), st, idx, end, rows, gp, cv)
    torch.cuda.synchronize()
    assert torch.isnan(cv).all() and np.isnan(r_cv).all()
    np.testing.assert_array_equal(idx.cpu().numpy(), r_idx)
    assert (idx.cpu().numpy() == 0).all()
    np.testing.assert_array_equal(end.cpu().numpy(), r_end)


@pytest.mark.parametrize("G", [7, 64, 200])
def test_goal_update_md_with_unnormalised_large_costs(dev, G):
    """cfg.normalize_cost = False with costs ~1e3: the mirror-descent exponents reach +-5e3.  The reference's termwise
    exp(L + z_j) stays finite near the root; a factored exp(L) * sum exp(z_j) is inf * 0 (found by tests/fuzz/fuzz_learner.py)."""
    from omg_planner_amd import _lib, ops
    from oracle import oracle as orc
    S, n = 3, 30
    rng = np.random.RandomState(G)
    traj, goals = rng.uniform(-2, 2, (S, n, 9)), rng.uniform(-2, 2, (S, G, 9))
    prm = _lib.LearnerParams()
    prm.alg, prm.num_goals, prm.n_waypoints, prm.start_idx = _lib.ALG["MD"], G, n, 3
    prm.constraint_num, prm.use_standoff, prm.normalize_cost = 1, 0, 0
    prm.base_obstacle_weight, prm.smooth_weight, prm.eta = 1.0, 0.01, float(np.sqrt(np.log(G + 1) / 50))
    po = orc.LearnerParams()
    for f, _ in po._fields_:
        setattr(po, f, getattr(prm, f))
    st_ref, st = orc.learner_state_init(S, G), ops.learner_state(S, G, dev)
    idx = torch.zeros(S, dtype=torch.int32, device=dev)
    end, rows, gp = (torch.zeros((S, 9), dtype=torch.float64, device=dev), torch.zeros((S, 1, 9), dtype=torch.float64, device=dev),
                     torch.zeros((S, 9), dtype=torch.float64, device=dev))
    for step in range(4):
        gc = rng.uniform(500, 1e3, (S, G)).astype(np.float32)
        gc[np.arange(S), rng.randint(0, G, S)] = 0.3  # one cheap goal per scene keeps the experts' costs (and q) finite
        r_idx, r_end, _, _, _ = orc.goal_update(po, traj, goals, None, gc, st_ref)
        ops.goal_update(prm, _t(traj, dev), _t(goals, dev), None, _t(gc, dev), st, idx, end, rows, gp)
        assert not np.isnan(st_ref).any() and not torch.isnan(st).any()
        np.testing.assert_array_equal(idx.cpu().numpy(), r_idx, err_msg=f"step {step}")
        np.testing.assert_allclose(st.cpu().numpy(), st_ref, rtol=1e-5, atol=1e-8, err_msg=f"step {step}")


def test_goal_collision_stats_match_oracle(dev):
    """Planner.setup_goal_set's collision filter (planner.py:512-524) for several scenes at once: per-goal collision counts
    (exact) and potential sums of the softened-finger layer against the oracle."""
    from omg_planner_amd import ops, robot as rb, scenes as sc
    from omg_planner_amd.goalset import goal_collision_stats, select_goals
    from oracle import oracle as orc
    S, G0 = 4, 90
    m = rb.PandaModel(seed=10)
    scenes, batch = _multi_scene_batch(S)
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G0, 50 + s) for s in range(S)])
    goals[:, ::3, :7] += np.random.RandomState(1).normal(0, 0.25, (S, len(range(0, G0, 3)), 7))  # some goals inside obstacles
    col, pot = goal_collision_stats(ops.robot_blob(m, dev), m.points_per_link, ops.DeviceScenes(batch, dev), _t(goals, dev))
    rp, _, rc = orc.fk_sdf(m.blob(), m.points_per_link, batch, goals, soften_fingers=True)
    np.testing.assert_array_equal(col.cpu().numpy(), rc.sum(axis=(-2, -1)))
    np.testing.assert_allclose(pot.cpu().numpy(), rp.sum(axis=(-2, -1)), rtol=1e-5, atol=1e-6)
    assert (rc.sum(axis=(-2, -1)) > 5).any() and (rc.sum(axis=(-2, -1)) <= 5).any()
    grasps, _, _, chosen = select_goals(list(goals[0]), None, col[0].cpu().numpy(), pot[0].cpu().numpy(), rng=np.random.RandomState(3))
    assert len(grasps) == len(chosen) > 0


# ------------------------------------------------------------------------------------------------
# (7) the whole planner loop: ChompEngine against the same loop driven through the oracle
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("alg,standoff", [("MD", False), ("FTL", True)])
def test_engine_plan_loop_matches_oracle_loop(dev, alg, standoff):
    """Planner.plan's loop body (omg/planner.py:612-621) for 4 scenes x 10 iterations: goal selection indices
    identical, trajectories within 1e-6 of the oracle-driven loop (every oracle piece is pinned to the reference)."""
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    from oracle import oracle as orc
    S, G, n, iters = 4, 6, 30, 10
    cfg = Config(timesteps=n, use_standoff=standoff, optim_steps=12)
    m = rb.PandaModel(seed=6)
    P = m.points_per_link
    scenes, batch = _multi_scene_batch(S)
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G, s) for s in range(S)])
    start = np.tile(rb.HOME_CONFIG, (S, 1))
    c = cfg.reach_tail_length if standoff else 1
    reach = None
    if standoff:
        reach = np.stack([[np.concatenate([sc.linear_init(g - np.array([0.1, -0.05, 0.1, 0.15, 0, -0.1, 0.1, 0, 0]), g, c - 1), g[None]], 0)
                           for g in goals[s]] for s in range(S)])
    eng = ChompEngine(m, batch, cfg, start, goals, reach_grasps=reach, device=dev, ol_alg=alg)
    traj = eng.traj.cpu().numpy().copy()
    state = orc.learner_state_init(S, G)
    cfg_o = Config(timesteps=n, use_standoff=standoff, optim_steps=12)
    cv_goals = reach[:, :, -1, :] if standoff else goals
    for t in range(iters):
        eng.iterate(t)
        # ---- oracle loop
        if t < cfg_o.optim_steps:
            lp = orc.LearnerParams()
            lp.alg, lp.num_goals, lp.n_waypoints = orc.ALG[alg], G, n
            lp.start_idx = min(int(((t + 1) / cfg_o.optim_steps) * n), n - 1)
            lp.constraint_num, lp.use_standoff, lp.normalize_cost = c, int(standoff), 1
            lp.base_obstacle_weight, lp.smooth_weight, lp.eta = 1.0, 0.1 * 0.1, float(np.sqrt(np.log(G + 1) / cfg_o.optim_steps))
            gc, _ = orc.goalset_cost(m.blob(), P, batch, traj[:, lp.start_idx], cv_goals, n - lp.start_idx, cfg_o.time_interval)
            idx, end, rows, gp, _ = orc.goal_update(lp, traj, goals, reach, gc, state)
        k = t + 1
        po = orc.ChompParams()
        src = eng._params(True)
        for f, _ in po._fields_:
            setattr(po, f, getattr(src, f))
        po.smoothness_weight = 0.1 * 1.02 ** k
        pot, pg, col = orc.fk_sdf(m.blob(), P, batch, traj)
        traj, _, _, info = orc.chomp_optimize(m.blob(), po, traj, start, end, rows, gp, pot, pg, col)
        np.testing.assert_array_equal(eng.goal_idx.cpu().numpy(), idx, err_msg=f"iteration {t}")
        np.testing.assert_allclose(eng.traj.cpu().numpy(), traj, rtol=0, atol=1e-6, err_msg=f"iteration {t}")
        np.testing.assert_allclose(eng.info.cpu().numpy()[:, :10], info[:, :10], rtol=1e-5, atol=1e-6, err_msg=f"iteration {t}")


@pytest.mark.parametrize("alg,standoff,n", [("MD", False, 30), ("Exp", True, 30), ("FTL", False, 64), ("Proj", False, 12)])
def test_fused_update_optimize_equals_separate_launches(dev, alg, standoff, n, monkeypatch):
    """omgx_goalset_cost_layer + omgx_goal_update_optimize == omgx_fk_sdf, omgx_goalset_cost, omgx_goal_update, omgx_chomp_optimize,
    bit for bit (n = 64: largest LDS layout; later iterations: goal-set window shorter than the trajectory)."""
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    S, G = 5, 9
    m = rb.PandaModel(seed=8)
    scenes, batch = _multi_scene_batch(S)
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G, s) for s in range(S)])
    start = np.tile(rb.HOME_CONFIG, (S, 1))
    cfg0 = Config()
    cfg0.use_standoff = standoff
    cfg0.optim_steps = 8
    cfg0.get_global_param(n)
    c = cfg0.reach_tail_length if standoff else 1
    reach = None
    if standoff:
        reach = np.stack([[np.concatenate([sc.linear_init(g - np.array([0.1, -0.05, 0.1, 0.15, 0, -0.1, 0.1, 0, 0]), g, c - 1), g[None]], 0)
                           for g in goals[s]] for s in range(S)])
    import copy
    outs = []
    for mode in ("split", "fused", "separate"):
        eng = ChompEngine(m, batch, copy.deepcopy(cfg0), start, goals, reach_grasps=reach, device=dev, ol_alg=alg)
        if mode == "split":    # two launches: goal-set batch + trajectory layer | learner and step in different workgroups
            eng.split_update = True
        elif mode == "fused":  # two launches, learner then step in one workgroup
            eng.split_update = False
        else:                  # five launches: trajectory layer, goal-set batch, goal update, step (iterate_separate)
            eng.separate_launches = True
        for t in range(5):
            eng.iterate(t)
        torch.cuda.synchronize()
        outs.append([x.clone() for x in (eng.traj, eng.info, eng.goal_idx, eng.learner_state, eng.end, eng.goal_rows, eng.cost_vec, eng.grad,
                                         eng.pot, eng.pgrad, eng.col, eng.goal_cost)])
    for other in outs[:-1]:
        for a, b in zip(other, outs[-1]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("case", ["md_switch_70", "exp_standoff_41", "md_early_2"])
@pytest.mark.parametrize("mode", ["fused", "serial", "latency", "split2", "split4"])
def test_engine_plan_matches_reference_planner_loop(dev, case, mode, monkeypatch):
    """ChompEngine on ONE scene against the reference's own planner run (tests/golden/plan_*.npz: Learner.__init__'s goal
    pick, then Learner.update_goal + Optimizer.optimize per iteration with the break on `terminate`, then the info-only
    evaluation; omg/planner.py:600-653), free-running for up to 70 iterations: same goal sequence, trajectories 1e-6."""
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    monkeypatch.setattr(ChompEngine, "separate_launches", mode == "serial")
    fx = H.load(f"plan_{case}.npz")
    m, batch = H.model_from(fx), H.batch_from(fx)
    standoff = bool(int(fx["cfg_use_standoff"]))
    cfg = Config(timesteps=30, use_standoff=standoff)
    assert cfg.optim_steps == int(fx["optim_steps"]) and cfg.extra_smooth_steps == int(fx["extra_smooth_steps"])
    eng = ChompEngine(m, batch, cfg, fx["start"][None], fx["goal_set"][None], reach_grasps=fx["reach_grasps"][None] if standoff else None,
                      device=dev, ol_alg=str(fx["alg"]), latency_mode=mode == "latency",  # latency: omgx_goalset_cost_layer_tiled
                      goal_parts=int(mode[5:]) if mode.startswith("split") else 1)     # split: omgx_goalset_cost_layer_parts
    eng.select_initial_goal()
    assert int(eng.goal_idx[0]) == int(fx["init_goal_idx"])
    np.testing.assert_allclose(eng.traj[0].cpu().numpy(), fx["init_traj"], rtol=0, atol=1e-12)
    iters = int(fx["iterations"])
    for t in range(cfg.optim_steps + cfg.extra_smooth_steps):
        eng.iterate(t, early_stop=True)
        if t < iters:
            assert int(eng.goal_idx[0]) == int(fx["selected_goals"][t]), t
            np.testing.assert_allclose(eng.traj[0].cpu().numpy(), fx["history"][t], rtol=0, atol=1e-6, err_msg=f"iteration {t}")
            info = eng.info[0].cpu().numpy()
            np.testing.assert_allclose(info[0], fx["info_cost"][t], rtol=1e-5, err_msg=f"iteration {t}")
            assert float(info[8]) == float(fx["info_collide"][t]) and bool(info[10] > 0.5) == bool(fx["info_terminate"][t]), t
        else:  # the reference has left its loop (planner.py:626); the engine's scene is inactive and keeps its trajectory
            assert int(eng.active[0]) == 0
            np.testing.assert_allclose(eng.traj[0].cpu().numpy(), fx["history"][iters - 1], rtol=0, atol=1e-6)
    assert bool(int(eng.active[0]) == 0) == bool(int(fx["terminated"]))
    if not int(fx["terminated"]):
        final = eng.optimize(False)[0].cpu().numpy()
        np.testing.assert_allclose(final[0], fx["info_cost"][-1], rtol=1e-5)
        np.testing.assert_allclose(final[2], fx["info_smooth"][-1], rtol=1e-6)


@pytest.mark.parametrize("case", ["md_switch_70", "exp_standoff_41", "md_early_2"])
def test_drop_in_classes_reproduce_reference_planner_loop(dev, case):
    """The single-scene drop-in level: Planner.plan's loop (omg/planner.py:600-653) written with the mirror classes
    Trajectory / Cost / Optimizer / Learner exactly as the reference writes it with its own, against the reference's run."""
    from omg_planner_amd.config import Config
    from omg_planner_amd.cost import Cost
    from omg_planner_amd.online_learner import Learner
    from omg_planner_amd.optimizer import Optimizer
    from omg_planner_amd.trajectory import Trajectory
    import types
    fx = H.load(f"plan_{case}.npz")
    standoff = bool(int(fx["cfg_use_standoff"]))
    cfg = Config(timesteps=30, use_standoff=standoff, ol_alg=str(fx["alg"]))
    env = _env_from(fx, dev, cfg)
    env.objects[env.target_idx].reach_grasps = fx["reach_grasps"]
    traj = Trajectory(cfg=cfg)
    traj.start, traj.goal_set, traj.end = fx["start"].copy(), fx["goal_set"], fx["goal_set"][0].copy()
    traj.interpolate_waypoints()
    cost = Cost(env)
    learner = Learner(env, traj, cost)
    optim = Optimizer(types.SimpleNamespace(config=cfg, robot=env.robot), cost)
    assert int(traj.goal_idx) == int(fx["init_goal_idx"])
    np.testing.assert_allclose(traj.data, fx["init_traj"], rtol=0, atol=1e-12)
    infos, history = [], []
    for t in range(cfg.optim_steps + cfg.extra_smooth_steps):
        if cfg.goal_set_proj and cfg.ol_alg not in ("Baseline", "Proj") and t < cfg.optim_steps:
            learner.update_goal()
        assert int(traj.goal_idx) == int(fx["selected_goals"][t]), t
        infos.append(optim.optimize(traj, force_update=True))
        history.append(np.copy(traj.data))
        np.testing.assert_allclose(traj.data, fx["history"][t], rtol=0, atol=1e-6, err_msg=f"iteration {t}")
        np.testing.assert_allclose(infos[-1]["cost"], fx["info_cost"][t], rtol=1e-5)
        if infos[-1]["terminate"] and t > 0:
            break
    assert len(history) == int(fx["iterations"]) and bool(infos[-1]["terminate"]) == bool(int(fx["terminated"]))
    if not infos[-1]["terminate"]:
        infos.append(optim.optimize(traj, info_only=True))
        np.testing.assert_allclose(infos[-1]["cost"], fx["info_cost"][-1], rtol=1e-5)
    assert abs(learner.p.sum() - 1.0) < 1e-7 and learner.t == min(len(history), cfg.optim_steps)  # Exp normalises with safe_div (+1e-8)


@pytest.mark.parametrize("case", ["md_switch_70", "exp_standoff_41"])
def test_drop_in_loop_gives_the_same_bits_however_it_is_driven(dev, case, monkeypatch):
    """device_loop.DeviceLoop (round 5): the planner loop through the drop-in classes (a) as omg/planner.py drives it — the goal index
    stored unread, the learner's update riding on optimize()'s fused launches —, (b) with the goal index READ right after
    update_goal() — the update then runs on its own and optimize() only steps —, (c) with Learner.DEFER_UPDATE = False, and (d) with
    host-side edits between the calls (traj.data replaced by an equal copy, the learner's distribution read every iteration): the
    same goals, the same trajectories bit for bit, the same info records; the learner's host attributes agree at the end."""
    from omg_planner_amd.config import Config
    from omg_planner_amd.cost import Cost
    from omg_planner_amd.device_loop import LazyIndex
    from omg_planner_amd.online_learner import Learner
    from omg_planner_amd.optimizer import Optimizer
    from omg_planner_amd.trajectory import Trajectory
    import types
    fx = H.load(f"plan_{case}.npz")
    standoff = bool(int(fx["cfg_use_standoff"]))
    iters = int(fx["iterations"])
    runs = {}
    for mode in ("planner", "read_at_once", "no_defer", "edits"):
        monkeypatch.setattr(Learner, "DEFER_UPDATE", mode != "no_defer")
        cfg = Config(timesteps=30, use_standoff=standoff, ol_alg=str(fx["alg"]))
        env = _env_from(fx, dev, cfg)
        env.objects[env.target_idx].reach_grasps = fx["reach_grasps"]
        traj = Trajectory(cfg=cfg)
        traj.start, traj.goal_set, traj.end = fx["start"].copy(), fx["goal_set"], fx["goal_set"][0].copy()
        traj.interpolate_waypoints()
        cost = Cost(env)
        learner = Learner(env, traj, cost)
        optim = Optimizer(types.SimpleNamespace(config=cfg, robot=env.robot), cost)
        goals, history, infos, lazy = [], [], [], 0
        for t in range(iters):
            if t < cfg.optim_steps:
                changed = learner.update_goal()
                lazy += isinstance(traj.goal_idx, LazyIndex) and not traj.goal_idx.resolved()
                if mode == "read_at_once":
                    assert int(traj.goal_idx) == int(fx["selected_goals"][t]) and isinstance(bool(changed), bool)
                if mode == "edits":
                    assert abs(float(np.sum(learner.p)) - 1.0) < 1e-6  # reading the distribution applies the update
                    traj.data = np.array(traj.data)                    # an equal copy: noticed, nothing to upload
                goals.append(traj.goal_idx)
            infos.append(optim.optimize(traj, force_update=True))
            history.append(np.copy(traj.data))
        infos.append(optim.optimize(traj, info_only=True))
        assert (lazy == min(iters, cfg.optim_steps)) == (mode in ("planner", "read_at_once", "edits")), (mode, lazy)
        runs[mode] = ([int(g) for g in goals], np.stack(history), np.array([[i["cost"], i["obs"], i["smooth"], i["grad"], float(i["terminate"])] for i in infos]),
                      np.array(learner.p), np.array(traj.end, np.float64))
    ref = runs["planner"]
    assert ref[0] == [int(g) for g in fx["selected_goals"][: len(ref[0])]]
    np.testing.assert_allclose(ref[1], fx["history"][:iters], rtol=0, atol=1e-6)
    for mode in ("read_at_once", "no_defer", "edits"):
        got = runs[mode]
        assert got[0] == ref[0], mode
        for a, b in zip(got[1:], ref[1:]):
            assert np.array_equal(a, b), mode


@pytest.mark.parametrize("mode", ["fused", "serial"])
def test_inactive_scenes_are_left_alone(dev, mode, monkeypatch):
    """Once a scene terminates the reference leaves its loop (omg/planner.py:626): no goal-set batch, no goal update, no
    step.  With an `active` mask the launches skip such scenes — every output of theirs (goal costs, layer, learner state,
    goal, trajectory, info) keeps its contents, and the other scenes' results are bit-identical to a run without any mask."""
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    import copy
    monkeypatch.setattr(ChompEngine, "separate_launches", mode == "serial")
    S, G, n = 11, 6, 30
    m = rb.PandaModel(seed=3)
    scenes, batch = _multi_scene_batch(S)
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G, s) for s in range(S)])
    start = np.tile(rb.HOME_CONFIG, (S, 1))
    cfg = Config(use_standoff=False)
    cfg.optim_steps = 6
    cfg.get_global_param(n)
    names = ("traj", "info", "goal_idx", "learner_state", "end", "goal_rows", "goal_point", "cost_vec", "grad", "pot", "pgrad", "col", "goal_cost")
    a = ChompEngine(m, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    b = ChompEngine(m, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    for t in range(2):
        a.iterate(t)
        b.iterate(t)
    mask = torch.tensor([1, 0, 1, 1, 0, 0, 1, 1, 1, 0, 1], dtype=torch.int32, device=dev)
    frozen = {k: getattr(b, k).clone() for k in names}
    b.active = mask.clone()
    for t in range(2, 5):
        a.iterate(t)
        b.iterate(t)
    torch.cuda.synchronize()
    on, off = mask.bool(), ~mask.bool()
    for k in names:
        assert torch.equal(getattr(b, k)[on], getattr(a, k)[on]), k       # the others: as without a mask
        if mode == "serial" and k in ("pot", "pgrad", "col", "goal_cost"):
            continue  # the separate entry points omgx_fk_sdf / omgx_goalset_cost take no mask: computed, then ignored
        assert torch.equal(getattr(b, k)[off], frozen[k][off]), k          # the masked ones: untouched since iteration 1
    assert not torch.equal(a.traj[off], frozen["traj"][off])


def test_plan_stops_launching_when_every_scene_has_terminated(dev):
    """ChompEngine.plan with early_stop looks at the active mask now and then and leaves the loop once nothing is active
    (planner.py:626 breaks at once): same results as running all 70 iterations over the inactive scenes."""
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    fx = H.load("plan_md_early_2.npz")  # the reference terminates after two iterations
    m, batch = H.model_from(fx), H.batch_from(fx)
    outs = []
    for stop_early in (True, False):
        eng = ChompEngine(m, batch, Config(timesteps=30, use_standoff=False), fx["start"][None], fx["goal_set"][None], device=dev, ol_alg="MD")
        if stop_early:
            info = eng.plan(early_stop=True)
            assert eng.iterations_run == 2 and int(eng.active[0]) == 0
        else:
            eng.select_initial_goal()
            for t in range(70):
                eng.iterate(t, early_stop=True)
            info = eng.optimize(False)
        outs.append((info.clone(), eng.traj.clone(), eng.goal_idx.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    np.testing.assert_allclose(outs[0][1][0].cpu().numpy(), fx["history"][1], rtol=0, atol=1e-6)
    np.testing.assert_allclose(outs[0][0][0, 0].item(), fx["info_cost"][-1], rtol=1e-5)


def test_engine_snapshot_restore_replays_identically(dev):
    """ChompEngine.snapshot / restore (bench.py restarts the plan with them): the same iterations after a restore give the
    same bits, including the learner state, the schedules and an active mask that early_stop had changed."""
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    S, G = 6, 5
    m = rb.PandaModel(seed=4)
    scenes, batch = _multi_scene_batch(S)
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G, s) for s in range(S)])
    cfg = Config(use_standoff=False)
    cfg.optim_steps = 4
    cfg.allow_collision_point = 10_000  # some scenes terminate: the active mask changes
    cfg.get_global_param(30)
    eng = ChompEngine(m, batch, cfg, np.tile(rb.HOME_CONFIG, (S, 1)), goals, device=dev, ol_alg="MD")
    eng.select_initial_goal()
    snap = eng.snapshot()
    runs = []
    for _ in range(2):
        for t in range(7):
            eng.iterate(t, early_stop=True)
        torch.cuda.synchronize()
        runs.append({k: getattr(eng, k).clone() for k in eng._STATE} | {"w": torch.tensor([cfg.smoothness_weight, float(eng.step_count), float(eng.t)])})
        eng.restore(snap)
    assert int(runs[0]["active"].sum()) < S
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k
    assert eng.step_count == 0 and eng.t == 0 and int(eng.active.sum()) == S


def test_goalset_slots_follow_the_active_scenes(dev):
    """omgx_goalset_cost_layer with a mask over 150 scenes (more than one 64-lane ballot): slot k of the grid works on the
    k-th active scene — outputs of active scenes equal the unmasked launch bit for bit, those of inactive scenes keep the
    sentinel they held; all-ones and all-zero masks included."""
    from omg_planner_amd import ops, robot as rb, scenes as sc
    S, G, n = 150, 3, 8
    m = rb.PandaModel(seed=6)
    P = m.points_per_link
    scenes = [sc.make_tabletop_scene(s % 7, grid=16, table_grid=(24, 16, 8)) for s in range(S)]
    batch = sc.pack_table(scenes, H.LAYER_CFG)
    rng = np.random.RandomState(3)
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G, s) for s in range(S)])
    traj = np.stack([sc.cubic_init(rb.HOME_CONFIG + rng.normal(0, 0.05, 9) * np.array([1] * 7 + [0, 0]), goals[s, 0], n) for s in range(S)])
    robot, ds = ops.robot_blob(m, dev), ops.DeviceScenes(batch, dev)
    tt, gg = _t(traj, dev), _t(goals, dev)

    def run(mask):
        lay = tuple(torch.full(shape, -7.0, dtype=torch.float32, device=dev) for shape in ((S, n, 10, P), (S, n, 10, P, 3), (S, n, 10, P)))
        out = (torch.full((S, G), -7.0, dtype=torch.float32, device=dev), torch.full((S, G), -7.0, dtype=torch.float32, device=dev))
        ops.goalset_cost_layer(robot, P, ds, tt[:, 2], gg, n - 2, 0.1, tt, lay, out=out,
                               active=None if mask is None else torch.as_tensor(mask, dtype=torch.int32, device=dev))
        torch.cuda.synchronize()
        return out + lay

    full = run(None)
    for mask in (np.ones(S, np.int32), (rng.rand(S) < 0.6).astype(np.int32), (np.arange(S) >= 140).astype(np.int32), np.zeros(S, np.int32)):
        got = run(mask)
        on = torch.as_tensor(mask.astype(bool), device=dev)
        for a, b in zip(got, full):
            assert torch.equal(a[on], b[on])
            assert bool((a[~on] == -7.0).all())


@pytest.mark.parametrize("alg,standoff,mode", [("MD", False, "fused"), ("Exp", True, "fused"), ("FTL", False, "serial"), ("Proj", False, "fused")])
def test_ragged_goal_sets_equal_single_scene_runs(dev, alg, standoff, mode, monkeypatch):
    """Scenes with different numbers of goals in ONE batch (goal_counts; arrays padded to the largest, padding filled with
    NaN so that any read of it would show): every scene's goal sequence, trajectory, info and learner state equal, bit for
    bit, those of the same scene planned alone with its own goal set."""
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    import copy
    monkeypatch.setattr(ChompEngine, "separate_launches", mode == "serial")
    counts = [3, 7, 1, 5, 7, 2]
    S, G, n = len(counts), max(counts), 30
    m = rb.PandaModel(seed=5)
    scenes, batch = _multi_scene_batch(S)
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G, 10 + s) for s in range(S)])
    start = np.tile(rb.HOME_CONFIG, (S, 1))
    cfg = Config(use_standoff=standoff)
    cfg.optim_steps = 5
    cfg.extra_smooth_steps = 2
    cfg.get_global_param(n)
    c = cfg.reach_tail_length
    reach = np.stack([[np.concatenate([sc.linear_init(g - np.array([0.1, -0.05, 0.1, 0.15, 0, -0.1, 0.1, 0, 0]), g, c - 1), g[None]], 0)
                       for g in goals[s]] for s in range(S)]) if standoff else None
    padded, padded_reach = goals.copy(), None if reach is None else reach.copy()
    for s, k in enumerate(counts):
        padded[s, k:] = np.nan
        if padded_reach is not None:
            padded_reach[s, k:] = np.nan
    eng = ChompEngine(m, batch, copy.deepcopy(cfg), start, padded, reach_grasps=padded_reach, device=dev, ol_alg=alg, goal_counts=counts)
    eng.select_initial_goal()
    singles = []
    for s, k in enumerate(counts):
        e1 = ChompEngine(m, batch.subset(s, s + 1), copy.deepcopy(cfg), start[s:s + 1], goals[s:s + 1, :k],
                         reach_grasps=None if reach is None else reach[s:s + 1, :k], device=dev, ol_alg=alg)
        e1.select_initial_goal()
        singles.append(e1)
    def same(a, b):  # bit-equal, NaN == NaN (a single-goal scene normalises its cost vector to 0/0 in the reference too)
        return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())

    for t in range(cfg.optim_steps + cfg.extra_smooth_steps):
        eng.iterate(t)
        for e1 in singles:
            e1.iterate(t)
        for s, (k, e1) in enumerate(zip(counts, singles)):
            assert int(eng.goal_idx[s]) == int(e1.goal_idx[0]) < k, (t, s)
            assert torch.equal(eng.traj[s], e1.traj[0]) and torch.equal(eng.info[s], e1.info[0]), (t, s)
            st, s1 = eng.learner_state[s], e1.learner_state[0]
            for blk in range(7):  # sum_costs | p | 5 x experts_p, padded stride G vs the single run's stride k
                assert same(st[blk * G: blk * G + k], s1[blk * k: (blk + 1) * k]), (t, s, blk)
            assert same(st[7 * G:], s1[7 * k:])
            assert torch.equal(eng.goal_cost[s, :k], e1.goal_cost[0]) or alg == "Proj"
    assert not torch.isnan(eng.info).any() and not torch.isnan(eng.traj).any()


def test_two_launch_entry_points_reject_bad_arguments(dev, monkeypatch):
    """omgx_goalset_cost_layer / omgx_goal_update_optimize: error codes, never a crash; odd sizes (1 scene, 1 goal, window
    shorter than the trajectory) agree with the separate entry points."""
    from omg_planner_amd import _lib, ops, robot as rb, scenes as sc
    import ctypes as C
    l = _lib.lib()
    rc = l.omgx_goalset_cost_layer(None, 15, None, None, None, None, 9, None, 1, 1, 5, 0.1, 0, None, None, None, None, 5, 0, None, None, None, None, None, None, 0, None, None)
    assert rc == _lib.OMGX_ERR_INVALID
    lp, cp = _lib.LearnerParams(), _lib.ChompParams()
    rc = l.omgx_goal_update_optimize(C.byref(lp), *([None] * 6), None, C.byref(cp), *([None] * 9), 3, *([None] * 4), None, 0, 0, None, None, None)
    assert rc == _lib.OMGX_ERR_INVALID
    # one scene, one goal, 7-waypoint window on a 12-waypoint trajectory
    m = rb.PandaModel(seed=2)
    P = m.points_per_link
    scenes, batch = _multi_scene_batch(1)
    n, G = 12, 1
    goals = np.stack([sc.make_reach_goals(scenes[0], m, G, 0)])
    traj = np.stack([sc.cubic_init(rb.HOME_CONFIG, goals[0, 0], n)])
    robot, ds = ops.robot_blob(m, dev), ops.DeviceScenes(batch, dev)
    tt, gg = _t(traj, dev), _t(goals, dev)
    lay = (torch.empty((1, n, 10, P), dtype=torch.float32, device=dev), torch.empty((1, n, 10, P, 3), dtype=torch.float32, device=dev),
           torch.empty((1, n, 10, P), dtype=torch.float32, device=dev))
    c1, k1 = ops.goalset_cost_layer(robot, P, ds, tt[:, 5], gg, n - 5, 0.1, tt, lay)
    c2, k2, _ = ops.goalset_cost(robot, P, ds, tt[:, 5], gg, n - 5, 0.1)
    # omgx_fk_sdf beyond OMGX_MAX_WAYPOINTS configurations takes its two-launch path (k_fk_poses + k_sdf_chunks<true>), a
    # trajectory-sized layer the layer workgroups of k_goalset_queue: same arithmetic, bit-identical outputs
    big = tt.repeat(1, 6, 1)[:, :70].contiguous()
    pb, gb, ob = ops.fk_sdf(robot, P, ds, big)
    p2, g2, o2 = pb[:, :n].contiguous(), gb[:, :n].contiguous(), ob[:, :n].contiguous()
    p3, g3, o3 = ops.fk_sdf(robot, P, ds, tt)
    assert torch.equal(p2, p3) and torch.equal(g2, g3) and torch.equal(o2, o3)
    assert torch.equal(c1, c2) and torch.equal(k1, k2)
    assert torch.equal(lay[0], p2.view_as(lay[0])) and torch.equal(lay[1], g2.view_as(lay[1])) and torch.equal(lay[2], o2.view_as(lay[2]))
    with pytest.raises(_lib.OmgHipError):  # layer outputs of the wrong size
        ops.goalset_cost_layer(robot, P, ds, tt[:, 5], gg, n - 5, 0.1, tt, (lay[0][:, :3], lay[1], lay[2]))


def test_split_update_with_more_scenes_than_compute_units(dev, monkeypatch):
    """600 workgroups of k_update_optimize_split cannot be resident at once on 256 CUs: the learner workgroups lead the
    grid, so a waiting optimiser workgroup always finds its producer dispatched.  Same results as the one-workgroup kernel."""
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    import copy
    S, G, n = 300, 4, 12
    m = rb.PandaModel(seed=9)
    scenes = [sc.make_tabletop_scene(s % 6, grid=24, table_grid=(32, 24, 8)) for s in range(S)]
    batch = sc.pack_table(scenes)
    rng = np.random.RandomState(5)
    goals = np.stack([sc.make_goal_set(s, G) for s in range(S)])
    start = np.tile(rb.HOME_CONFIG, (S, 1)) + rng.normal(0, 0.02, (S, 9)) * np.array([1] * 7 + [0, 0])
    cfg0 = Config()
    cfg0.use_standoff = False
    cfg0.optim_steps = 6
    cfg0.get_global_param(n)
    outs = []
    for split in (True, False):
        eng = ChompEngine(m, batch, copy.deepcopy(cfg0), start, goals, device=dev, ol_alg="MD")
        eng.split_update = split  # the engine would not split this many scenes by itself
        for t in range(3):
            eng.iterate(t)
        torch.cuda.synchronize()
        outs.append([x.clone() for x in (eng.traj, eng.info, eng.goal_idx, eng.learner_state)])
    assert not torch.isnan(outs[0][1]).any()
    for a, b in zip(*outs):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------
# (8) BASELINE config 5 shape (kitchen-like: 50 waypoints, 12 obstacle SDFs incl. a point-cloud SDF) and
#     full-size properties at BASELINE config 4 size (100 scenes x 128 goals x 30 waypoints)
# ------------------------------------------------------------------------------------------------
def _kitchen_scene(seed):
    from omg_planner_amd import scenes as sc
    rng = np.random.RandomState(seed)
    objs = []
    for i in range(9):
        g = (24, 20, 16)[i % 3]
        sdf = sc.sphere_sdf(rng.uniform(0.05, 0.09), (g, g, g), 0.5 / g) if i % 2 else sc.box_sdf(rng.uniform(0.03, 0.1, 3), (g, g, g), 0.5 / g)
        objs.append(sc.SceneObject(f"obj_{i}", sc._yaw_pose(rng.uniform(0.2, 0.8), rng.uniform(-0.4, 0.4), rng.uniform(0.1, 0.6), rng.uniform(-3, 3)), sdf))
    cloud = rng.uniform([0.3, -0.3, 0.0], [0.7, 0.3, 0.4], size=(4096, 3))  # PointEnv.compute_sdf_from_points input
    objs.append(sc.SceneObject("perception/env_points", np.eye(4), sc.point_cloud_sdf(cloud)))
    objs.append(sc.SceneObject("floor", sc._yaw_pose(0, 0, -0.2, 0), sc.box_sdf((0.4, 0.4, 0.02), (12, 12, 8), 0.1)))
    objs.append(sc.SceneObject("table", sc._yaw_pose(0.5, 0, 0.02, 0), sc.box_sdf((0.6, 0.4, 0.02), (48, 36, 12), 0.03)))
    return sc.Scene(objs, target_idx=1)


def test_kitchen_config_50_waypoints_12_objects_matches_oracle(dev):
    from omg_planner_amd import _lib, ops, robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from oracle import oracle as orc
    S, n, G = 3, 50, 5
    cfg = Config(use_standoff=False)
    cfg.get_global_param(n)  # called from the 30-step default state: dt = 0.1 * 30 / 50 = 0.06 (omg/config.py:201)
    assert abs(cfg.time_interval - 0.06) < 1e-12 and cfg.timesteps == n
    m = rb.PandaModel(seed=8)
    P = m.points_per_link
    scenes = [_kitchen_scene(s) for s in range(S)]
    batch = sc.pack_table(scenes, cfg.layer_kwargs())
    assert all(batch.scene_begin[s + 1] - batch.scene_begin[s] == 12 for s in range(S))
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G, s) for s in range(S)])
    start = np.tile(rb.HOME_CONFIG, (S, 1))
    traj = np.stack([sc.cubic_init(start[s], goals[s, 0], n) for s in range(S)])
    robot, ds = ops.robot_blob(m, dev), ops.DeviceScenes(batch, dev)
    # goal-set cost with a 37-waypoint window
    gc_ref, col_ref = orc.goalset_cost(m.blob(), P, batch, traj[:, 13], goals, 37, cfg.time_interval)
    gc, col, _ = ops.goalset_cost(robot, P, ds, _t(traj[:, 13], dev), _t(goals, dev), 37, cfg.time_interval)
    np.testing.assert_allclose(gc.cpu().numpy(), gc_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(col.cpu().numpy(), col_ref)
    # optimiser steps
    fx = dict(cfg_top_k=1000, cfg_goal_set_proj=1, cfg_use_standoff=0, cfg_dt=cfg.time_interval)
    t_dev, t_ref = _t(traj, dev), traj.copy()
    for step in (1, 2, 3):
        po = H.params_from(fx, orc.ChompParams, n, P, 1, 1.0, 0.1 * 1.02 ** step)
        pd = H.params_from(fx, _lib.ChompParams, n, P, 1, 1.0, 0.1 * 1.02 ** step)
        rp, rg, rc = orc.fk_sdf(m.blob(), P, batch, t_ref)
        t_ref, g_ref, _, i_ref = orc.chomp_optimize(m.blob(), po, t_ref, start, goals[:, 0], goals[:, :1], goals[:, 0], rp, rg, rc)
        pot, pg, cl = ops.fk_sdf(robot, P, ds, t_dev)
        assert (pot.cpu().numpy() == rp).mean() > 0.999
        g, _, info = ops.chomp_optimize(robot, pd, t_dev, _t(start, dev), _t(goals[:, 0], dev), _t(goals[:, :1], dev),
                                        _t(goals[:, 0], dev), pot, pg, cl)
        np.testing.assert_allclose(t_dev.cpu().numpy(), t_ref, rtol=0, atol=1e-7)
        np.testing.assert_allclose(info.cpu().numpy()[:, :10], i_ref[:, :10], rtol=1e-6, atol=1e-6)
    assert i_ref[:, 8].max() > 0  # the scene really collides somewhere


def test_goalset_full_size_properties(dev):
    """BASELINE config 4 size on one GPU: 100 scenes x 128 goals x 30 waypoints = 57.6 M points per launch.
    Size-independent properties: run-to-run identical (no atomics), duplicated goals get identical costs,
    permuting the goals permutes the costs, the per-goal cost equals the in-order sum of its potentials."""
    from omg_planner_amd import ops, robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    S, G, n = 100, 128, 30
    cfg = Config(timesteps=n, use_standoff=False)
    m = rb.PandaModel(seed=0)
    P = m.points_per_link
    scenes = [sc.make_tabletop_scene(s, grid=32, table_grid=(64, 48, 16)) for s in range(S)]
    batch = sc.pack_table(scenes, cfg.layer_kwargs())
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G, s) for s in range(S)])
    goals[:, 7] = goals[:, 3]  # a duplicated goal
    start = np.tile(rb.HOME_CONFIG, (S, 1))
    ts = np.stack([sc.cubic_init(start[s], goals[s, 0], n)[0] for s in range(S)])
    robot, ds = ops.robot_blob(m, dev), ops.DeviceScenes(batch, dev)
    c1, k1, _ = ops.goalset_cost(robot, P, ds, _t(ts, dev), _t(goals, dev), n, 0.1)
    c1, k1 = c1.clone(), k1.clone()
    c2, k2, _ = ops.goalset_cost(robot, P, ds, _t(ts, dev), _t(goals, dev), n, 0.1)
    assert torch.equal(c1, c2) and torch.equal(k1, k2)
    assert torch.equal(c1[:, 7], c1[:, 3])
    perm = np.random.RandomState(0).permutation(G)
    c3, _, _ = ops.goalset_cost(robot, P, ds, _t(ts, dev), _t(goals[:, perm], dev), n, 0.1)
    assert torch.equal(c3, c1[:, torch.from_numpy(perm).to(dev)])
    # potentials of a slice of scenes: per-goal cost == sum over its [n,10,P] potentials (float32, loose order tolerance)
    sub = sc.SceneBatch(batch.objects[: batch.scene_begin[8]].copy(), batch.scene_begin[:9].copy(), batch.pool)
    c4, _, pots = ops.goalset_cost(robot, P, ops.DeviceScenes(sub, dev), _t(ts[:8], dev), _t(goals[:8], dev), n, 0.1, want_potentials=True)
    # cost-only launch (k_goalset_queue: float32 sum of pot * weight in queue order) against the per-point path
    # (k_sdf_chunks: per-point sums over objects, then the weight): the same pairs, a different float32 summation order
    np.testing.assert_allclose(c4.cpu().numpy(), c1[:8].cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pots.double().sum(dim=(2, 3, 4)).cpu().numpy(), c4.double().cpu().numpy(), rtol=2e-6, atol=1e-6)
    assert float((c1 > 0).float().mean()) > 0.9


# ------------------------------------------------------------------------------------------------
# (9) omgx_point_cloud_sdf — PointEnv.compute_sdf_from_points (BASELINE config 5 input path)
# ------------------------------------------------------------------------------------------------
def test_point_cloud_sdf_matches_oracle_and_feeds_the_sdf_layer(dev):
    from omg_planner_amd import ops, scenes as sc
    from oracle import oracle as orc
    from tests.test_oracle_pointcloud import reference_grid
    rng = np.random.RandomState(3)
    pts = rng.uniform([0.3, -0.3, 0.0], [0.7, 0.3, 0.4], size=(4096, 3))
    ref, origin, dims = reference_grid(pts)  # scipy cKDTree, as the reference calls it
    grid, org, res = ops.point_cloud_sdf(_t(pts, dev))
    torch.cuda.synchronize()
    g = grid.cpu().numpy()
    assert g.shape == ref.shape and np.array_equal(org, origin)
    np.testing.assert_array_equal(g, orc.point_cloud_sdf(pts, origin, 0.02, dims))  # same arithmetic: bit-exact vs oracle
    assert (g == ref).mean() > 0.9999
    np.testing.assert_allclose(g, ref, rtol=0, atol=1e-7)
    # the device-built grid is a valid obstacle for the SDF layer: same potentials as with the host-built grid
    host = sc.point_cloud_sdf(pts)
    np.testing.assert_allclose(host.data, ref, rtol=0, atol=0)
    q = np.random.RandomState(4).uniform([0.2, -0.4, 0.0], [0.8, 0.4, 0.5], size=(20000, 3)).astype(np.float32)
    def layer(data):
        scene = sc.Scene([sc.SceneObject("perception/env_points", np.eye(4), sc.SdfGrid(data, origin, 0.02))], 0)
        sdf, lim = sc.pack_padded(scene.objects)
        poses, eps, pad, clr, dis = sc.layer_params(scene, **H.LAYER_CFG)
        return ops.sdf_loss_forward(*[_t(a, dev) for a in (poses, sdf, lim, q, eps, pad, clr, dis)])
    a, b = layer(g), layer(ref)
    assert float((a[0] > 0).float().mean()) > 0.3
    np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), rtol=0, atol=1e-6)


def test_cost_forward_kinematics_obstacle_matches_reference_fixture(dev):
    """The reference's intermediate tensors (x, v, a, per-link Jacobians) from Cost.forward_kinematics_obstacle."""
    from omg_planner_amd.cost import Cost
    from omg_planner_amd.util import wrap_joint
    for case in ("topk1000", "short_n5"):
        fx = H.load(f"cost_{case}.npz")
        n = fx["xi"].shape[0]
        cost = Cost(_env_from(fx, dev, _cfg_from(fx, n)))
        x, v, a, Js, pot, pgrad, vis, col = cost.forward_kinematics_obstacle(fx["xi"], fx["start"], fx["end"])
        np.testing.assert_allclose(x, fx["x"], atol=1e-12)
        np.testing.assert_allclose(v, fx["v"], atol=1e-9)
        np.testing.assert_allclose(a, fx["a"], atol=1e-7)
        for j in range(10):
            k = len(wrap_joint(j + 1))
            np.testing.assert_allclose(Js[j][..., :3], fx["J"][:, j, :, :k], atol=1e-12)
        np.testing.assert_allclose(pot, fx["potentials"], atol=2e-6)
        assert float(col) == float(fx["collide_sum"])
        assert vis.shape == (n, 10, pot.shape[2], 12)
        # Cost.compute_obstacle_cost_layer on explicit points (the op through the reference's own call shape, cost.py:288-360)
        pts = torch.as_tensor(x.astype(np.float32), device=dev)
        vis2 = np.zeros_like(vis)
        p2, g2, c2 = cost.compute_obstacle_cost_layer(pts, vis2, special_check_id=0, uncheck_finger_collision=int(fx["cfg_uncheck"]))
        np.testing.assert_allclose(p2.cpu().numpy(), fx["potentials"], atol=2e-6)
        np.testing.assert_allclose(g2.cpu().numpy(), fx["potential_grads"], atol=2e-4)
        assert float(c2.sum()) == float(fx["collide_sum"])
        np.testing.assert_array_equal(vis2[..., :3], x.astype(np.float32))
        np.testing.assert_array_equal(vis2[..., 6], p2.cpu().numpy())
        # only_collide (cost.py:279-284): the whole batch is kept or zeroed by ONE any() over it
        q = np.concatenate([fx["xi"], fx["xi"][::-1]], 0)
        pa, _, _, _ = cost.batch_obstacle_cost(q, only_collide=False, uncheck_finger_collision=0, want_vis=False)
        pb, _, _, _ = cost.batch_obstacle_cost(q, only_collide=True, uncheck_finger_collision=0, want_vis=False)
        thr = 0.5 * (cost.cfg.epsilon - cost.cfg.clearance) ** 2 / cost.cfg.epsilon
        assert torch.equal(pb, pa * bool((pa > thr).any()))


def test_maximum_waypoints_64_matches_oracle(dev):
    """The build's size limits (64 waypoints, 16 points per link, 8 constraint rows) against the oracle."""
    from omg_planner_amd import _lib, ops, robot as rb, scenes as sc
    from oracle import oracle as orc
    S, n, G, Pn, c = 2, 64, 3, 16, 8
    m = rb.PandaModel(points_per_link=Pn, seed=9)
    scenes, batch = _multi_scene_batch(S)
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G, s) for s in range(S)])
    start = np.tile(rb.HOME_CONFIG, (S, 1))
    traj = np.stack([sc.cubic_init(start[s], goals[s, 0], n) for s in range(S)])
    robot, ds = ops.robot_blob(m, dev), ops.DeviceScenes(batch, dev)
    gc_ref, _ = orc.goalset_cost(m.blob(), Pn, batch, traj[:, 0], goals, n, 0.05)
    gc, _, _ = ops.goalset_cost(robot, Pn, ds, _t(traj[:, 0], dev), _t(goals, dev), n, 0.05)
    np.testing.assert_allclose(gc.cpu().numpy(), gc_ref, rtol=1e-5, atol=1e-6)
    reach = np.stack([[np.concatenate([sc.linear_init(g - 0.1, g, c - 1), g[None]], 0) for g in goals[s]] for s in range(S)])
    fx = dict(cfg_top_k=1000, cfg_goal_set_proj=1, cfg_use_standoff=1, cfg_dt=0.05)
    po = H.params_from(fx, orc.ChompParams, n, Pn, 1, 1.0, 0.102, reach_tail_length=c)
    pd = H.params_from(fx, _lib.ChompParams, n, Pn, 1, 1.0, 0.102, reach_tail_length=c)
    rp, rg, rc = orc.fk_sdf(m.blob(), Pn, batch, traj)
    t_ref, g_ref, _, i_ref = orc.chomp_optimize(m.blob(), po, traj, start, goals[:, 0], reach[:, 0], goals[:, 0], rp, rg, rc)
    t_dev = _t(traj, dev)
    pot, pg, cl = ops.fk_sdf(robot, Pn, ds, t_dev)
    g, _, info = ops.chomp_optimize(robot, pd, t_dev, _t(start, dev), _t(goals[:, 0], dev), _t(reach[:, 0], dev), _t(goals[:, 0], dev), pot, pg, cl)
    np.testing.assert_allclose(t_dev.cpu().numpy(), t_ref, rtol=0, atol=1e-7)
    np.testing.assert_allclose(g.cpu().numpy(), g_ref, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(info.cpu().numpy()[:, :10], i_ref[:, :10], rtol=1e-6, atol=1e-6)
    # one past the limits is refused with an error code, not a crash
    pd.n_waypoints = 65
    with pytest.raises(_lib.OmgHipError):
        ops.chomp_optimize(robot, pd, t_dev, _t(start, dev), _t(goals[:, 0], dev), _t(reach[:, 0], dev), _t(goals[:, 0], dev), pot, pg, cl)


def test_engine_full_plan_runs_and_improves(dev):
    """Planner.plan through ChompEngine (initial goal pick, 50 + 20 iterations, final info): costs finite, the
    goal-set constraint holds (last waypoint on the selected goal) and the smoothness cost dropped."""
    from omg_planner_amd import robot as rb, scenes as sc
    from omg_planner_amd.config import Config
    from omg_planner_amd.engine import ChompEngine
    S, G, n = 6, 16, 30
    cfg = Config(timesteps=n, use_standoff=False)
    m = rb.PandaModel(seed=0)
    scenes = [sc.make_tabletop_scene(s, grid=32, table_grid=(64, 48, 16)) for s in range(S)]
    batch = sc.pack_table(scenes, cfg.layer_kwargs())
    goals = np.stack([sc.make_reach_goals(scenes[s], m, G, s) for s in range(S)])
    eng = ChompEngine(m, batch, cfg, np.tile(rb.HOME_CONFIG, (S, 1)), goals, device=dev, ol_alg="MD")
    first = eng.optimize(False).clone()
    eng.step_count = 0
    info = eng.plan(early_stop=False)
    torch.cuda.synchronize()
    assert torch.isfinite(info).all() and torch.isfinite(eng.traj).all()
    idx = eng.goal_idx.long()
    sel = eng.goal_set[torch.arange(S, device=dev), idx]
    assert float((eng.traj[:, -1] - sel).abs().max()) < 1e-9       # goal_set_projection pins the end point
    assert float(info[:, 9].max()) < 1e-9                           # info["reach"]
    lo = torch.as_tensor(m.joint_lower_limit, device=dev) - 1e-2
    hi = torch.as_tensor(m.joint_upper_limit, device=dev) + 1e-2
    assert bool(((eng.traj >= lo) & (eng.traj <= hi)).all())       # handle_joint_limit
    assert float(info[:, 8].sum()) <= float(first[:, 8].sum())     # fewer colliding points than the initial guess


# ------------------------------------------------------------------------------------------------
# (10) the timed workload itself: bench.py's configuration against the oracle, and its multi-rank path on one GPU
# ------------------------------------------------------------------------------------------------
def test_bench_workload_matches_oracle(dev):
    """bench.py's exact configuration — 100 scenes (4 x 64^3 + 128x96x32 private grids), 64 goals, 30 waypoints, MD, the
    two-launch iteration with the goal-set window pinned, schedule measured on the second launch — against the oracle on three
    of its scenes: first through the measuring launches, then three more iterations under the measured schedule."""
    import bench
    from omg_planner_amd.engine import ChompEngine
    from oracle.check import engine_vs_oracle
    cfg, model, batch, start, goals = bench.build_workload(100, 64, 30, 64, 0, False)
    eng = ChompEngine(model, batch, cfg, start, goals, device=dev, ol_alg="MD")
    for phase in range(2):
        r = engine_vs_oracle(eng, batch, [0, 50, 99], steps=3, pin_window=True)
        assert r["goal_idx_equal"], r
        assert r["max_traj_err"] <= 1e-6 and r["max_cost_rel_err"] <= 1e-5, r  # north_star's bar is 1e-4
    assert eng._measured and eng.schedule is not None


def test_bench_multi_rank_costs_equal_single_process(tmp_path):
    """bench.py --total-scenes (strong scaling: contiguous blocks of whole scenes per rank, one all-gather of the final
    costs) with 2 ranks sharing the one GPU over gloo against the single-process run: the gathered per-scene costs are
    bit-identical.  The ranks are fresh processes started by torch.distributed.run before they touch the GPU."""
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parents[1]
    common = ["--total-scenes", "6", "--goals", "8", "--grid", "24", "--steps", "4", "--warmup", "1", "--no-plan", "--no-cpu-baseline",
              "--no-parity"]
    env = dict(os.environ, OMGX_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    one, two = tmp_path / "one.npy", tmp_path / "two.npy"
    # (--layout-scenes 3: the single process lays its engine out like a rank of the 2-rank job, see ChompEngine.layout)
    r1 = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", *common, "--layout-scenes", "3", "--dump-costs", str(one)], env=env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    port = 29500 + os.getpid() % 2000
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", *common, "--dump-costs", str(two)], env=env,
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    assert a.shape == (6,) and np.array_equal(a, b), (a, b)
    line = [l for l in r2.stdout.splitlines() if l.startswith("{")][-1]
    import json
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["total_scenes"] == 6


def test_bench_two_ranks_on_the_config4_share_shape(tmp_path):
    """BASELINE config 4's per-rank shape: 25 scenes x 128 goals over 2 ranks = shards of 13 + 12 scenes (what ranks of the
    8-GPU job hold), launched by torch.distributed.run with 2 ranks on the one GPU over gloo.  Every rank picks its layout from
    the LARGEST shard (ChompEngine.layout(13, 128)), so the two shards — and the single-process run told the same number — compute
    the same bits: gathered per-scene costs bit-identical; the line carries a complete per-rank roofline list."""
    import json
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parents[1]
    common = ["--total-scenes", "25", "--goals", "128", "--steps", "6", "--warmup", "2", "--no-plan", "--no-cpu-baseline", "--no-parity"]
    env = dict(os.environ, OMGX_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    one, two = tmp_path / "one.npy", tmp_path / "two.npy"
    r1 = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "1", *common, "--layout-scenes", "13", "--dump-costs", str(one)], env=env,
                        capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    port = 31500 + os.getpid() % 2000
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", *common, "--dump-costs", str(two)], env=env,
                        capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    a, b = np.load(one), np.load(two)
    assert a.shape == (25,) and np.isfinite(a).all() and np.array_equal(a, b), (a, b)
    j1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    j = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["total_scenes"] == 25 and j["steps"] == 6
    assert j["config"]["layout"]["evaluated_for_scenes"] == 13 and j1["config"]["layout"]["evaluated_for_scenes"] == 13
    assert j["config"]["layout"]["goal_parts"] == j1["config"]["layout"]["goal_parts"]
    pr = j["roofline"]["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and [r["scenes"] for r in pr] == [13, 12]
    for r in pr:
        assert set(r) >= {"rank", "scenes", "ms_per_step", "avg_launch_ms", "launches", "achieved", "frac", "hbm_GBs", "algorithmic_equiv_GBs"}
        assert r["launches"] > 0 and r["avg_launch_ms"] > 0 and r["ms_per_step"] > 0 and r["algorithmic_equiv_GBs"] > 0
        # since round 5 profiles/roofline_inputs.json holds the counts of this shape too (13 x 128, three pipeline parts; the shard
        # of 12 scenes takes them scaled by 12 / 13): every rank's achieved rate and fraction is a number
        assert r["achieved"] is not None and 0.0 < r["frac"] <= 1.0, r
    assert j["roofline"]["profiled_workload"]["goals"] == 128 and j["roofline"]["profiled_workload"]["scenes"] == 13
    assert j["roofline"]["counts_scaled"] == pytest.approx(1.0) and j1["roofline"]["frac"] is not None
    assert "cpu_baseline" not in j  # an N = 1 field
    # the timed regions: `steps` is what was asked for, ms_per_step the median region, the spread beside it
    for line in (j, j1):
        lo, hi = line["ms_per_step_spread"]
        assert 5 <= line["regions"] <= 15 and lo <= line["ms_per_step"] <= hi and line["steps"] == 6
        assert abs(line["value"] - 25 * 6 / (line["ms_per_step"] * 6e-3)) <= 1e-6 * line["value"]
        assert line["setup_ms"]["pack_table_ms"] > 0 and line["setup_ms"]["engine_init_ms"] > 0
