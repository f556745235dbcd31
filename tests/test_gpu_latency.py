"""Latency mode (ChompEngine(latency_mode=True), omgx_goalset_cost_layer_tiled): one or a few scenes — BASELINE configs 1-2, the
shape of the reference's own Planner.plan (omg/planner.py:600-653: one scene, one plan) — with the goal-set batch and the trajectory
layer cut into many small workgroups.  What has to hold: layer outputs bit for bit whatever the split; a goal's cost = the
float32 sum of its parts' sums, within summation rounding of the batch layout and at the oracle's tolerance; plans like the batch
layout's and like the oracle's; the error paths of the new entry point."""
from __future__ import annotations

import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _workload(S, G, grid=32, seed=0, n=30):
    import bench
    return bench.build_workload(S, G, n, grid, seed, False)


def _make(dev, S, G, latency, counts=None, grid=32, alg="MD", n=30):
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = _workload(S, G, grid, n=n)
    return ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=alg, goal_counts=counts, latency_mode=latency), batch


@pytest.mark.parametrize("lg,cb", [(1, 0), (2, 7), (5, 16), (10, 16), (10, 4)])
def test_layer_outputs_do_not_depend_on_the_split(dev, lg, cb):
    """Every element of the trajectory layer is computed on its own: 10 / lg links x cb waypoints per workgroup, spread or not,
    against omgx_fk_sdf's five workgroups per scene."""
    from omg_planner_amd import ops
    eng, _ = _make(dev, 3, 8, False)
    ref = ops.fk_sdf(eng.robot, eng.P, eng.scenes, eng.traj)
    for spread in (False, True):
        out = tuple(torch.full_like(t, float("nan")) for t in ref)
        ops.goalset_cost_layer_tiled(eng.robot, eng.P, eng.scenes, None, None, 1, eng.cfg.time_interval, eng.traj, out, None,
                                     layer_link_groups=lg, layer_config_block=cb, spread=spread, goal_parts=1)
        torch.cuda.synchronize()
        for a, b in zip(ref, out):
            assert torch.equal(a, b), (lg, cb, spread)


@pytest.mark.parametrize("parts", [1, 2, 4, 8])
@pytest.mark.parametrize("n_rem", [30, 17, 5, 1])
def test_goal_cost_parts_add_up_to_the_batch_cost_and_the_oracle(dev, parts, n_rem):
    """[S][G][NP] partial sums: their float32 sum in part order against the batch layout (another summation order: 1e-6) and the
    oracle (1e-5, the bar of the batch layout); collision counts are integers and add up exactly."""
    from omg_planner_amd import ops
    from oracle import oracle as orc
    eng, batch = _make(dev, 2, 24, False)
    ts = eng.traj[:, 30 - n_rem]
    cost, col, _ = ops.goalset_cost(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval)
    NP = ops.goalset_parts(n_rem, parts)
    assert 1 <= NP <= parts and (NP == 1 or ((n_rem + 3) // 4) * 5 // NP >= 4)
    pc = torch.full((2, 24 * NP), float("nan"), dtype=torch.float32, device=dev)
    pl = torch.full_like(pc, float("nan"))
    got = ops.goalset_cost_layer_tiled(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, None, None, (pc, pl),
                                       goal_parts=parts, spread=True)
    assert got == NP
    torch.cuda.synchronize()
    tot = pc.reshape(2, 24, NP)[:, :, 0].clone()
    for k in range(1, NP):
        tot += pc.reshape(2, 24, NP)[:, :, k]
    np.testing.assert_allclose(tot.cpu().numpy(), cost.cpu().numpy(), rtol=2e-6, atol=1e-7)
    assert torch.equal(pl.reshape(2, 24, NP).sum(-1), col)
    for s in range(2):
        gc, _ = orc.goalset_cost(eng.model.blob(), eng.P, batch.subset(s, s + 1), ts[s:s + 1].cpu().numpy(), eng.cv_goals[s:s + 1].cpu().numpy(),
                                 n_rem, eng.cfg.time_interval)
        np.testing.assert_allclose(tot[s].cpu().numpy(), np.asarray(gc).reshape(-1), rtol=1e-5, atol=1e-6)
    if parts == 1:  # one part per goal and a scene per XCD: the tiled entry point launches the batch kernel — the same bits
        bc = torch.full((2, 24), float("nan"), dtype=torch.float32, device=dev)
        bl = torch.full_like(bc, float("nan"))
        ops.goalset_cost_layer_tiled(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, None, None, (bc, bl),
                                     goal_parts=1, spread=False)
        torch.cuda.synchronize()
        assert torch.equal(bc, cost) and torch.equal(bl, col)


def test_latency_engine_follows_the_batch_engine_and_the_oracle(dev):
    """Three scenes, ragged goal sets: iterations of a plan in both modes — layer outputs equal bit for bit, goal costs within
    summation rounding, the same goals chosen, trajectories equal (the learner's distribution only enters through its arg-max) —
    and the latency engine against the oracle-driven loop."""
    from oracle.check import engine_vs_oracle
    counts = np.array([20, 13, 7])
    a, _ = _make(dev, 3, 20, False, counts)
    b, _ = _make(dev, 3, 20, True, counts)
    for e in (a, b):
        e.select_initial_goal()
    assert torch.equal(a.goal_idx, b.goal_idx) and torch.equal(a.traj, b.traj)
    for t in (0, 1, 2, 20, 35, 49, 50, 55):
        for e in (a, b):
            e.t = t
            e.iterate(t, early_stop=t > 1)
        torch.cuda.synchronize()
        for k in ("pot", "pgrad", "col"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (t, k)
        if t < 50:
            ga, gb = a.goal_cost_total().cpu().numpy(), b.goal_cost_total().cpu().numpy()
            for s in range(3):
                np.testing.assert_allclose(gb[s, :counts[s]], ga[s, :counts[s]], rtol=2e-6, atol=1e-7)
        assert torch.equal(a.goal_idx, b.goal_idx), t
        np.testing.assert_allclose(b.traj.cpu().numpy(), a.traj.cpu().numpy(), rtol=0, atol=1e-9)
    c, batch = _make(dev, 2, 16, True)
    c.select_initial_goal()
    r = engine_vs_oracle(c, batch, [0, 1], steps=12, pin_window=False)
    assert r["goal_idx_equal"] and r["max_traj_err"] <= 1e-6 and r["max_cost_rel_err"] <= 1e-5, r


@pytest.mark.parametrize("early", [True, False])
def test_latency_plan_equals_the_batch_plan_and_its_graph(dev, early):
    """A whole plan of one scene x 64 goals (BASELINE config 2) in latency mode: the same goals and trajectories as the batch
    layout (1e-9), and captured as one HIP graph it replays to the bits of plan()."""
    a, _ = _make(dev, 1, 64, False, grid=64)
    b, _ = _make(dev, 1, 64, True, grid=64)
    a.plan(early_stop=early)
    b.plan(early_stop=early)
    torch.cuda.synchronize()
    assert torch.equal(a.goal_idx, b.goal_idx) and torch.equal(a.active, b.active)
    np.testing.assert_allclose(b.traj.cpu().numpy(), a.traj.cpu().numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(b.info.cpu().numpy(), a.info.cpu().numpy(), rtol=1e-9, atol=1e-12)
    c, _ = _make(dev, 1, 64, True, grid=64)
    fresh = c.snapshot()
    graph = c.capture_plan(early_stop=early)
    c.restore(fresh)
    graph.replay()
    torch.cuda.synchronize()
    for k in ("traj", "info", "goal_idx", "learner_state", "end", "goal_rows"):
        assert np.array_equal(getattr(c, k).cpu().numpy(), getattr(b, k).cpu().numpy(), equal_nan=True), k
    assert c._pipeline_parts() == 1 and c.schedule is None  # no pipeline, no dispatch schedule in this mode


def test_tiled_entry_point_rejects_bad_arguments(dev):
    import ctypes as C
    from omg_planner_amd import _lib, ops
    eng, _ = _make(dev, 1, 8, False)
    l = _lib.lib()
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    cost = torch.zeros(8 * 8, dtype=torch.float32, device=dev)

    def call(goal_parts=4, lg=10, cb=16, spread=1, goals=eng.cv_goals, traj=eng.traj, G=8):
        return l.omgx_goalset_cost_layer_tiled(p(eng.robot), eng.P, p(eng.scenes.objects), p(eng.scenes.scene_begin), p(eng.scenes.pool),
                                               p(eng.traj), 270, p(goals), 1, G, 30, 0.1, 0, p(cost), p(cost), p(traj), 30, 0,
                                               p(eng.pot), p(eng.pgrad), p(eng.col), None, None, goal_parts, lg, cb, spread, None, None, None)
    assert call() == _lib.OMGX_OK
    assert call(goal_parts=0) == _lib.OMGX_ERR_INVALID and call(goal_parts=9) == _lib.OMGX_ERR_INVALID
    assert call(lg=3) == _lib.OMGX_ERR_INVALID and call(lg=0) == _lib.OMGX_ERR_INVALID and call(cb=-1) == _lib.OMGX_ERR_INVALID
    assert call(goal_parts=4, spread=0) == _lib.OMGX_OK  # since ABI 8: the batch kernel with split goals (tests/test_gpu_parts.py)
    assert call(goal_parts=1, spread=0) == _lib.OMGX_OK
    assert call(goals=None, traj=None, G=0) == _lib.OMGX_ERR_INVALID   # nothing to do
    assert call(goals=None, G=8) == _lib.OMGX_ERR_INVALID
    assert l.omgx_goalset_parts(30, 4) == 4 and l.omgx_goalset_parts(30, 8) == 8 and l.omgx_goalset_parts(12, 8) == 2
    assert l.omgx_goalset_parts(4, 8) == 1 and l.omgx_goalset_parts(0, 4) == 0 and l.omgx_goalset_parts(30, 9) == 0
    torch.cuda.synchronize()
    with pytest.raises(_lib.OmgHipError):  # the partial sums need S * G * parts elements
        ops.goalset_cost_layer_tiled(eng.robot, eng.P, eng.scenes, eng.traj[:, 0], eng.cv_goals, 30, 0.1, None, None,
                                     (cost[:8], cost[:8]), goal_parts=4)


@pytest.mark.parametrize("alg,split", [("MD", None), ("FTL", False), ("Proj", None)])
def test_pose_hand_over_changes_no_bit(dev, alg, split, monkeypatch):
    """Inside a latency-mode plan the launches hand link poses to each other (ABI 7: the layer workgroups' waypoint poses, the
    tabulated start / goal poses) instead of running the same kinematics again: every result bit for bit what the kernels compute
    on their own — learner and step in two workgroups or in one, early stop, ragged goal sets."""
    from omg_planner_amd.engine import ChompEngine
    counts = np.array([16, 9, 12])
    out = []
    for on in (True, False):
        monkeypatch.setattr(ChompEngine, "LAT_HAND_OVER_POSES", on)
        e, _ = _make(dev, 3, 16, True, counts, alg=alg)
        e.split_update = split
        e.plan(early_stop=True)
        torch.cuda.synchronize()
        assert not e._poses_on
        out.append({k: getattr(e, k).cpu().numpy().copy() for k in ("traj", "info", "goal_idx", "learner_state", "grad", "cost_traj", "end", "goal_rows")})
        out[-1]["active"] = e.active.cpu().numpy().copy()
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k], equal_nan=True), k


@pytest.mark.parametrize("n", [6, 13])
@pytest.mark.parametrize("alg", ["MD", "FTL"])
def test_pose_hand_over_below_14_waypoints_keeps_the_end_pose_current(dev, n, alg, monkeypatch):
    """Below 14 waypoints the end pose does not fit the grad rows the split update launch hands it over in (end_pose_fits): the
    learner's workgroup must still keep `end_pose` on the chosen goal, because the plan's later launches — the smoothing
    iterations and the final evaluation (k_chomp_optimize) — read it.  A plan whose goal changes on the way, learner and step in
    two workgroups: every bit as without the hand-over."""
    from omg_planner_amd.engine import ChompEngine
    out = []
    for on in (True, False):
        monkeypatch.setattr(ChompEngine, "LAT_HAND_OVER_POSES", on)
        e, _ = _make(dev, 2, 16, True, alg=alg, n=n)
        e.split_update = True
        first = None
        e.select_initial_goal()
        first = e.goal_idx.cpu().numpy().copy()
        e.plan(early_stop=False, initial_goal=False)
        torch.cuda.synchronize()
        out.append({k: getattr(e, k).cpu().numpy().copy() for k in ("traj", "info", "goal_idx", "grad", "cost_traj", "end", "goal_rows")})
        out[-1]["first"] = first
        if on:  # the end pose the plan left behind is the chosen goal's
            from omg_planner_amd import ops
            want = ops.pose_table(e.robot, e.P, e.end).cpu().numpy()
            assert np.array_equal(e.end_pose.cpu().numpy(), want)
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k], equal_nan=True), k


def test_pose_table_is_the_forward_kinematics_before_the_centre_offset(dev):
    """omgx_pose_table against omgx_forward_kinematics (pinned to the reference's robot_pykdl by tests/golden/fk.npz): the 4 x 4
    link pose is the tabulated pose times the link's centre offset (robot_pykdl.py:203-204)."""
    from omg_planner_amd import ops
    from omg_planner_amd.robot import PandaModel
    m = PandaModel(points_per_link=15)
    robot = ops.robot_blob(m, dev)
    rng = np.random.RandomState(3)
    q = torch.as_tensor(rng.uniform(-2.5, 2.5, (37, 9)) * np.array([1] * 7 + [0.016, 0.016]), dtype=torch.float64, device=dev).contiguous()
    tab = ops.pose_table(robot, 15, q).cpu().numpy()           # [37,10,12]
    full, _, _ = ops.forward_kinematics(robot, 15, q, want_joint_info=False)
    full = full.cpu().numpy()                                    # [37,10,4,4]
    co = m.center_offset.reshape(10, 4, 4)
    P4 = np.zeros((37, 10, 4, 4))
    P4[:, :, :3, :3] = tab[:, :, :9].reshape(37, 10, 3, 3)
    P4[:, :, :3, 3] = tab[:, :, 9:]
    P4[:, :, 3, 3] = 1.0
    np.testing.assert_allclose(P4 @ co[None], full, rtol=0, atol=1e-12)


def test_latency_mode_at_the_waypoint_limit(dev):
    """64 waypoints (OMGX_MAX_WAYPOINTS): the latency-mode kernel then needs 73 KB of LDS (poses of 65 configurations + the chain
    constants + the layer's per-object contributions) and opts in beyond the 64 KB default; a window of 64 configurations has
    16 x 5 tiles.  Against the batch layout on the same scenes."""
    import bench
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(2, 24, 64, 32, 1, False)
    mk = lambda lat: ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", latency_mode=lat)
    a, b = mk(False), mk(True)
    for e in (a, b):
        e.select_initial_goal()
    assert torch.equal(a.goal_idx, b.goal_idx)
    for t in (0, 1, 30, 49, 50):
        for e in (a, b):
            e.t = t
            e.iterate(t)
        torch.cuda.synchronize()
        for k in ("pot", "pgrad", "col"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (t, k)
        if t < 50:
            np.testing.assert_allclose(b.goal_cost_total().cpu().numpy(), a.goal_cost_total().cpu().numpy(), rtol=2e-6, atol=1e-7)
        assert torch.equal(a.goal_idx, b.goal_idx), t
        np.testing.assert_allclose(b.traj.cpu().numpy(), a.traj.cpu().numpy(), rtol=0, atol=1e-9)
    c = mk(True)
    c.plan(early_stop=True)  # with the pose hand-over
    d = mk(False)
    d.plan(early_stop=True)
    torch.cuda.synchronize()
    assert torch.equal(c.goal_idx, d.goal_idx)
    np.testing.assert_allclose(c.traj.cpu().numpy(), d.traj.cpu().numpy(), rtol=0, atol=1e-9)


def test_latency_mode_with_a_dozen_objects(dev):
    """BASELINE config 5's shape (50 waypoints, 12 obstacle volumes + table): the layer's object-parallel evaluation walks the
    objects in groups of 8 (two rounds here), the scalar-cache warm-up covers the first 8 records only."""
    import bench
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(2, 16, 50, 32, 2, False, num_objects=12)
    mk = lambda lat: ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", latency_mode=lat)
    a, b = mk(False), mk(True)
    for e in (a, b):
        e.select_initial_goal()
    for t in (0, 3, 40, 50, 51):
        for e in (a, b):
            e.t = t
            e.iterate(t)
        torch.cuda.synchronize()
        for k in ("pot", "pgrad", "col"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (t, k)
        if t < 50:
            np.testing.assert_allclose(b.goal_cost_total().cpu().numpy(), a.goal_cost_total().cpu().numpy(), rtol=2e-6, atol=1e-7)
        assert torch.equal(a.goal_idx, b.goal_idx), t
        np.testing.assert_allclose(b.traj.cpu().numpy(), a.traj.cpu().numpy(), rtol=0, atol=1e-9)
