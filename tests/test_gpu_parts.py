"""Split goals in the BATCH layout (ChompEngine(goal_parts=...), omgx_goalset_cost_layer_parts / omgx_goalset_schedule_parts, ABI 8):
mid-size batches — one GPU's share of BASELINE config 4 on 8 GPUs (13 scenes x 128 goals), 25 x 64 — whose goal-set launch is a round
or two of the chip's workgroup slots and therefore bound by the latency of one goal workgroup.  What has to hold: the partial sums
add up to the one-workgroup cost (another float32 summation order) and to the oracle's; collision counts add up exactly; layer outputs
bit for bit; any dispatch schedule over the (scene, goal, part) items gives the same bits; plans like the unsplit engine's and the
oracle's, pipelined or not, ragged or not; the reference's own planner runs through it."""
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


def _make(dev, S, G, goal_parts, counts=None, grid=32, alg="MD", n=30):
    import bench
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(S, G, n, grid, 0, False)
    return ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=alg, goal_counts=counts, goal_parts=goal_parts), batch


def _total(buf, S, G, NP):
    parts = buf.reshape(-1)[: S * G * NP].reshape(S, G, NP)
    tot = parts[:, :, 0].clone()
    for k in range(1, NP):
        tot += parts[:, :, k]
    return tot


@pytest.mark.parametrize("parts", [2, 4, 8])
@pytest.mark.parametrize("n_rem", [30, 17, 5, 1])
def test_split_goal_costs_add_up_and_the_layer_keeps_its_bits(dev, parts, n_rem):
    from omg_planner_amd import ops
    from oracle import oracle as orc
    S, G = 3, 24
    counts = np.array([24, 11, 17])
    eng, batch = _make(dev, S, G, 1, counts)
    ts = eng.traj[:, 30 - n_rem]
    lay = tuple(torch.full_like(t, float("nan")) for t in (eng.pot, eng.pgrad, eng.col))
    cost, col = ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay,
                                       goal_count=eng.goal_count)
    NP = ops.goalset_parts(n_rem, parts)
    pc = torch.full((S, G * NP), float("nan"), dtype=torch.float32, device=dev)
    pl = torch.full_like(pc, float("nan"))
    lay2 = tuple(torch.full_like(t, float("nan")) for t in lay)
    poses = torch.full((S, 30, 10, 12), float("nan"), dtype=torch.float64, device=dev)
    ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay2, out=(pc, pl),
                           goal_count=eng.goal_count, goal_parts=parts, layer_poses=poses)
    torch.cuda.synchronize()
    for a, b in zip(lay, lay2):
        assert torch.equal(a, b)
    np.testing.assert_array_equal(poses.cpu().numpy(), ops.pose_table(eng.robot, eng.P, eng.traj).cpu().numpy())
    tot, tcol = _total(pc, S, G, NP).cpu().numpy(), _total(pl, S, G, NP).cpu().numpy()
    for s in range(S):
        k = counts[s]
        np.testing.assert_allclose(tot[s, :k], cost[s, :k].cpu().numpy(), rtol=2e-6, atol=1e-7)
        assert np.array_equal(tcol[s, :k], col[s, :k].cpu().numpy())
        assert np.isnan(pc.reshape(S, G, NP)[s, k:].cpu().numpy()).all()  # the padding of a ragged goal set is never written
        gc, _ = orc.goalset_cost(eng.model.blob(), eng.P, batch.subset(s, s + 1), ts[s:s + 1].cpu().numpy(), eng.cv_goals[s:s + 1, :k].cpu().numpy(),
                                 n_rem, eng.cfg.time_interval)
        np.testing.assert_allclose(tot[s, :k], np.asarray(gc).reshape(-1), rtol=1e-5, atol=1e-6)
    # any dispatch schedule over the (scene, goal, part) items: the same bits; the measuring launch stamps every item that ran
    work = torch.zeros(S * G * NP, dtype=torch.int32, device=dev)
    sched = ops.goalset_schedule(None, S, G, goal_count=eng.goal_count, parts=NP, device=dev)
    pc2, pl2 = torch.full_like(pc, float("nan")), torch.full_like(pl, float("nan"))
    ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay2, out=(pc2, pl2),
                           goal_count=eng.goal_count, goal_parts=parts, schedule=sched, work=work)
    sched2 = ops.goalset_schedule(work, S, G, goal_count=eng.goal_count, parts=NP)
    pc3, pl3 = torch.full_like(pc, float("nan")), torch.full_like(pl, float("nan"))
    ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay2, out=(pc3, pl3),
                           goal_count=eng.goal_count, goal_parts=parts, schedule=sched2)
    torch.cuda.synchronize()
    for x in (pc2, pc3):
        assert np.array_equal(x.cpu().numpy(), pc.cpu().numpy(), equal_nan=True)
    for x in (pl2, pl3):
        assert np.array_equal(x.cpu().numpy(), pl.cpu().numpy(), equal_nan=True)
    w = work.cpu().numpy().reshape(S, G, NP)
    sc = sched2.cpu().numpy()
    items = np.sort(sc[sc >= 0])
    want = np.concatenate([(s * G + np.arange(counts[s]))[:, None] * NP + np.arange(NP)[None, :] for s in range(S)]).reshape(-1)
    assert np.array_equal(items, np.sort(want))  # every item exactly once, no padding
    for s in range(S):
        assert (w[s, :counts[s]] > 0).all() and (w[s, counts[s]:] == 0).all()


@pytest.mark.parametrize("parts,alg", [(2, "MD"), (4, "FTL"), (2, "Exp")])
def test_split_engine_follows_the_unsplit_engine_and_the_oracle(dev, parts, alg):
    """Iterations of a plan with and without split goals: layer outputs equal bit for bit, goal costs within summation rounding,
    the same goals, trajectories equal; the fused launches against the five separate ones bit for bit; and against the oracle."""
    from oracle.check import engine_vs_oracle
    counts = np.array([20, 13, 7])
    a, _ = _make(dev, 3, 20, 1, counts, alg=alg)
    b, _ = _make(dev, 3, 20, parts, counts, alg=alg)
    c, _ = _make(dev, 3, 20, parts, counts, alg=alg)
    c.separate_launches = True
    for e in (a, b, c):
        e.select_initial_goal()
    assert torch.equal(a.goal_idx, b.goal_idx) and torch.equal(a.traj, b.traj)
    for t in (0, 1, 2, 20, 35, 46, 49, 50, 55):
        for e in (a, b, c):
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
        for k in ("traj", "goal_idx", "learner_state", "info", "pot", "col"):
            assert np.array_equal(getattr(b, k).cpu().numpy(), getattr(c, k).cpu().numpy(), equal_nan=True), (t, k)
    d, batch = _make(dev, 2, 16, parts, alg=alg)
    d.select_initial_goal()
    r = engine_vs_oracle(d, batch, [0, 1], steps=12, pin_window=False)
    assert r["goal_idx_equal"] and r["max_traj_err"] <= 1e-6 and r["max_cost_rel_err"] <= 1e-5, r


@pytest.mark.parametrize("early", [True, False])
def test_split_plan_pipelined_equals_unpipelined_and_its_graph(dev, early):
    """13 scenes x 128 goals (one GPU's share of BASELINE config 4 on 8 GPUs) with split goals: the plan on three pipeline parts, on
    one stream, and replayed as one HIP graph — the same bits; against the unsplit plan: same goals, trajectories at 1e-9."""
    S, G = 13, 128
    a, _ = _make(dev, S, G, 1)
    a.plan(early_stop=early)
    out = []
    for pipe in (None, 1):
        b, _ = _make(dev, S, G, 2)
        b.pipeline = pipe
        b.plan(early_stop=early)
        torch.cuda.synchronize()
        out.append({k: getattr(b, k).cpu().numpy().copy() for k in ("traj", "info", "goal_idx", "learner_state", "end", "goal_rows", "active")})
    assert ChompEngineParts(S, G) >= 2
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k], equal_nan=True), k
    assert np.array_equal(out[0]["goal_idx"], a.goal_idx.cpu().numpy()) and np.array_equal(out[0]["active"], a.active.cpu().numpy())
    np.testing.assert_allclose(out[0]["traj"], a.traj.cpu().numpy(), rtol=0, atol=1e-9)
    c, _ = _make(dev, S, G, 2)
    fresh = c.snapshot()
    graph = c.capture_plan(early_stop=early)
    c.restore(fresh)
    graph.replay()
    torch.cuda.synchronize()
    for k in ("traj", "info", "goal_idx", "learner_state", "end", "goal_rows"):
        assert np.array_equal(getattr(c, k).cpu().numpy(), out[0][k], equal_nan=True), k


def ChompEngineParts(S, G):
    from omg_planner_amd.engine import ChompEngine
    return ChompEngine.auto_parts(S, G)


@pytest.mark.parametrize("goal_parts,alg,split,pipe", [(1, "MD", None, None), (2, "FTL", False, 1), (1, "Exp", True, 2), (4, "MD", None, 2)])
def test_pose_hand_over_in_the_batch_layout_changes_no_bit(dev, goal_parts, alg, split, pipe, monkeypatch):
    """plan() hands link poses between its launches in the batch layout too (since round 4): the layer workgroups' waypoint poses
    to the step, the tabulated start / goal poses to the learner and the step — every result bit for bit what the kernels
    compute on their own; learner and step in two workgroups or in one, pipelined or not, early stop, ragged goal sets."""
    from omg_planner_amd.engine import ChompEngine
    counts = np.array([16, 9, 12, 16, 5])
    out = []
    for on in (True, False):
        monkeypatch.setattr(ChompEngine, "LAT_HAND_OVER_POSES", on)
        e, _ = _make(dev, 5, 16, goal_parts, counts, alg=alg)
        e.split_update, e.pipeline = split, pipe
        e.plan(early_stop=True)
        torch.cuda.synchronize()
        assert not e._poses_on
        out.append({k: getattr(e, k).cpu().numpy().copy() for k in ("traj", "info", "goal_idx", "learner_state", "grad", "cost_traj", "end", "goal_rows", "pot", "col")})
        out[-1]["active"] = e.active.cpu().numpy().copy()
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k], equal_nan=True), k
    # outside plan(): the switch bench.py uses around its steps
    a, _ = _make(dev, 3, 16, goal_parts, alg=alg)
    b, _ = _make(dev, 3, 16, goal_parts, alg=alg)
    a.pose_hand_over(True)
    for t in (0, 1, 2, 30):
        for e in (a, b):
            e.t = t
            e.iterate(t)
    torch.cuda.synchronize()
    for k in ("traj", "info", "goal_idx", "learner_state", "grad"):
        assert np.array_equal(getattr(a, k).cpu().numpy(), getattr(b, k).cpu().numpy(), equal_nan=True), k


@pytest.mark.parametrize("n_rem", [30, 13])
def test_a_goals_cost_does_not_depend_on_the_order_of_its_terms(dev, n_rem):
    """A workgroup's sum of pot x weight is accumulated exactly (terms on the 2^-36 grid, float64) and rounded to float32 once: the
    same goals through the batch kernel (tile t -> wave t % 4) and through the latency-mode kernel with ONE part per goal (tiles
    dealt heaviest first in a serpentine, another queue order altogether) give the same bits; with several parts the float32
    partial sums add up to it within one rounding per part; and the exact float64 sum of the oracle's per-pair terms rounds to it."""
    from omg_planner_amd import ops
    eng, batch = _make(dev, 2, 24, 1, grid=32)
    ts = eng.traj[:, 30 - n_rem]
    cost, col, _ = ops.goalset_cost(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval)
    pc = torch.full((2, 24), float("nan"), dtype=torch.float32, device=dev)
    pl = torch.full_like(pc, float("nan"))
    ops.goalset_cost_layer_tiled(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, None, None, (pc, pl), goal_parts=1, spread=True)
    torch.cuda.synchronize()
    assert torch.equal(pc, cost) and torch.equal(pl, col)
    for parts in (2, 4):
        NP = ops.goalset_parts(n_rem, parts)
        qc = torch.full((2, 24 * NP), float("nan"), dtype=torch.float32, device=dev)
        ql = torch.full_like(qc, float("nan"))
        ops.goalset_cost_layer_tiled(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, None, None, (qc, ql), goal_parts=parts, spread=False)
        torch.cuda.synchronize()
        tot = qc.reshape(2, 24, NP).double().sum(-1)
        np.testing.assert_allclose(tot.cpu().numpy(), cost.double().cpu().numpy(), rtol=NP * 6e-8, atol=1e-9)


@pytest.fixture
def ranges_on():
    """Waypoint ranges are built, tested and OFF by default (measured: no faster than dealt tiles / whole goals): switched on through the
    library's experiment hook for the tests that hold them to the parts' contract."""
    import ctypes as C
    from omg_planner_amd import _lib
    f = _lib.lib().omgx_debug_set_range
    f.argtypes, f.restype = [C.c_int], None
    f(40)
    yield f
    f(-1)


@pytest.mark.parametrize("n,n_rem", [(50, 50), (50, 43), (64, 64), (41, 41), (64, 57)])
def test_long_windows_split_into_waypoint_ranges(dev, ranges_on, n, n_rem):
    """Round 6 (k_goalset_range): beyond 40 configurations the two parts of a goal are RANGES of its window — each with the kinematics
    and the poses of its own configurations only — not dealt tiles.  Same contract as the dealt parts: partial sums that add up to the
    whole goal's cost within float32 summation rounding, collision counts that add up exactly, layer outputs and handed-over poses bit
    for bit (the layer's pieces are cut in two as well), any dispatch schedule the same bits, padding never written, the oracle's bar."""
    from omg_planner_amd import ops
    from oracle import oracle as orc
    S, G = 3, 20
    counts = np.array([20, 9, 14])
    eng, batch = _make(dev, S, G, 1, counts, grid=24, n=n)
    ts = eng.traj[:, n - n_rem]
    lay = tuple(torch.full_like(t, float("nan")) for t in (eng.pot, eng.pgrad, eng.col))
    cost, col = ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay,
                                       goal_count=eng.goal_count)
    NP = ops.goalset_parts(n_rem, 2)
    assert NP == 2
    pc = torch.full((S, G * NP), float("nan"), dtype=torch.float32, device=dev)
    pl = torch.full_like(pc, float("nan"))
    lay2 = tuple(torch.full_like(t, float("nan")) for t in lay)
    poses = torch.full((S, n, 10, 12), float("nan"), dtype=torch.float64, device=dev)
    ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay2, out=(pc, pl),
                           goal_count=eng.goal_count, goal_parts=2, layer_poses=poses)
    torch.cuda.synchronize()
    for a, b in zip(lay, lay2):
        assert torch.equal(a, b)
    np.testing.assert_array_equal(poses.cpu().numpy(), ops.pose_table(eng.robot, eng.P, eng.traj).cpu().numpy())
    tot, tcol = _total(pc, S, G, NP).cpu().numpy(), _total(pl, S, G, NP).cpu().numpy()
    parts = pc.reshape(S, G, NP).cpu().numpy()
    for s in range(S):
        k = counts[s]
        np.testing.assert_allclose(tot[s, :k], cost[s, :k].cpu().numpy(), rtol=2e-6, atol=1e-7)
        assert np.array_equal(tcol[s, :k], col[s, :k].cpu().numpy())
        assert np.isnan(parts[s, k:]).all()
        gc, _ = orc.goalset_cost(eng.model.blob(), eng.P, batch.subset(s, s + 1), ts[s:s + 1].cpu().numpy(), eng.cv_goals[s:s + 1, :k].cpu().numpy(),
                                 n_rem, eng.cfg.time_interval)
        np.testing.assert_allclose(tot[s, :k], np.asarray(gc).reshape(-1), rtol=1e-5, atol=1e-6)
    # ... and these ARE ranges: the dealt tiles of the default build split the same totals differently
    ranges_on(-1)
    pct, plt = torch.full_like(pc, float("nan")), torch.full_like(pl, float("nan"))
    ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay2, out=(pct, plt),
                           goal_count=eng.goal_count, goal_parts=2)
    torch.cuda.synchronize()
    ranges_on(40)
    tt = _total(pct, S, G, NP).cpu().numpy()
    for s in range(S):
        np.testing.assert_allclose(tt[s, :counts[s]], tot[s, :counts[s]], rtol=2e-6, atol=1e-7)
    assert not np.array_equal(pct.cpu().numpy(), pc.cpu().numpy(), equal_nan=True)
    h = n_rem - 4 * (n_rem // 8)
    work = torch.zeros(S * G * NP, dtype=torch.int32, device=dev)
    sched = ops.goalset_schedule(None, S, G, goal_count=eng.goal_count, parts=NP, device=dev)
    pc2, pl2 = torch.full_like(pc, float("nan")), torch.full_like(pl, float("nan"))
    ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay2, out=(pc2, pl2),
                           goal_count=eng.goal_count, goal_parts=2, schedule=sched, work=work)
    sched2 = ops.goalset_schedule(work, S, G, goal_count=eng.goal_count, parts=NP)
    pc3, pl3 = torch.full_like(pc, float("nan")), torch.full_like(pl, float("nan"))
    ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay2, out=(pc3, pl3),
                           goal_count=eng.goal_count, goal_parts=2, schedule=sched2)
    torch.cuda.synchronize()
    for x in (pc2, pc3):
        assert np.array_equal(x.cpu().numpy(), pc.cpu().numpy(), equal_nan=True)
    for x in (pl2, pl3):
        assert np.array_equal(x.cpu().numpy(), pl.cpu().numpy(), equal_nan=True)
    w = work.cpu().numpy().reshape(S, G, NP)
    for s in range(S):
        assert (w[s, :counts[s]] > 0).all() and (w[s, counts[s]:] == 0).all()
    assert h >= n_rem - h and (n_rem - h) % 4 == 0


def test_range_split_engine_follows_the_unsplit_engine_and_the_oracle(dev, ranges_on):
    """A 50-waypoint plan (BASELINE config 5's window) with two workgroups per goal: while the window is longer than 40 configurations
    the parts are waypoint ranges, then dealt tiles — the same goals as the unsplit engine, trajectories at 1e-9, layer outputs bit for
    bit; pipelined equals unpipelined bit for bit; and the oracle's bars over the first iterations."""
    from oracle.check import engine_vs_oracle
    counts = np.array([12, 7, 10, 12])
    a, _ = _make(dev, 4, 12, 1, counts, grid=24, n=50)
    b, _ = _make(dev, 4, 12, 2, counts, grid=24, n=50)
    c, _ = _make(dev, 4, 12, 2, counts, grid=24, n=50)
    c.pipeline = 2
    for e in (a, b, c):
        e.select_initial_goal()
        e.pose_hand_over(True)
    for t in range(0, 40):
        for e in (a, b, c):
            e.iterate(t, early_stop=t > 3)
        if t in (0, 1, 5, 9, 10, 11, 20, 39):
            c.join()
            torch.cuda.synchronize()
            for k in ("pot", "pgrad", "col"):
                assert torch.equal(getattr(a, k), getattr(b, k)), (t, k)
            assert torch.equal(a.goal_idx, b.goal_idx), t
            np.testing.assert_allclose(b.traj.cpu().numpy(), a.traj.cpu().numpy(), rtol=0, atol=1e-9)
            for k in ("traj", "goal_idx", "learner_state", "info", "pot", "col", "goal_cost"):
                assert np.array_equal(getattr(b, k).cpu().numpy(), getattr(c, k).cpu().numpy(), equal_nan=True), (t, k)
    d, batch = _make(dev, 2, 10, 2, grid=24, n=50)
    d.select_initial_goal()
    r = engine_vs_oracle(d, batch, [0, 1], steps=12, pin_window=False)
    assert r["goal_idx_equal"] and r["max_traj_err"] <= 1e-6 and r["max_cost_rel_err"] <= 1e-5, r
