"""ChompEngine's software pipeline (two scene ranges on two HIP streams, engine.py `_iterate_pipelined`) must not change a
bit: the parts run the same launches on row views of the engine's tensors.  Scenes are independent in the reference
(omg/core.py:869-885 plans them one after the other), so there is nothing to compare but the engine with itself — and the
pipelined engine with the oracle at bench.py's configuration.
"""
from __future__ import annotations

import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

STATE = ("traj", "info", "goal_idx", "learner_state", "goal_cost", "end", "goal_rows", "cost_traj", "grad", "pot", "col")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _engines(dev, S, G, ragged=False, grid=32):
    import bench
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(S, G, 30, grid, 0, False)
    counts = None
    if ragged:
        counts = np.random.RandomState(5).randint(G // 2, G + 1, S)
    make = lambda: ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", goal_counts=counts)
    return make, batch


def _assert_same(a, b):
    torch.cuda.synchronize()
    for k in STATE:
        x, y = getattr(a, k).cpu().numpy(), getattr(b, k).cpu().numpy()
        assert np.array_equal(x, y, equal_nan=True), k
    assert np.array_equal(a.active.cpu().numpy(), b.active.cpu().numpy())


@pytest.mark.parametrize("parts,ragged", [(2, False), (3, True)])
def test_pipelined_iterations_equal_the_single_stream_engine(dev, parts, ragged):
    make, _ = _engines(dev, 70, 64, ragged)
    one, two = make(), make()
    two.pipeline = parts
    for t in range(6):  # through the uniform schedule, the measuring launch and the measured schedule of every part
        one.iterate(t)
        two.iterate(t)
    assert two._parts is not None and len(two._parts) == parts
    assert all(p._measured for p in two._parts)  # every part above ChompEngine.MEASURE_MIN_ITEMS (256) items has measured its own schedule
    _assert_same(one, two)
    # whole-batch operations join the side streams by themselves
    snap = two.snapshot()
    two.iterate(6)
    two.restore(snap)
    one_costs = one.final_costs().cpu().numpy()
    assert np.array_equal(two.final_costs().cpu().numpy(), one_costs)
    assert (one.t, one.step_count, one.cfg.obstacle_weight, one.cfg.step_size) == (two.t, two.step_count, two.cfg.obstacle_weight, two.cfg.step_size)


def test_pipelined_plan_with_early_stop_equals_the_single_stream_plan(dev):
    make, _ = _engines(dev, 64, 64)
    one, two = make(), make()
    one.pipeline = 1
    assert two.pipeline is None and two.auto_parts(64, 64) == 3 and two.auto_parts(400, 64) == 2 and two.auto_parts(13, 128) == 3 and two.auto_parts(8, 64) == 1  # plan() decides
    i1 = one.plan(early_stop=True).cpu().numpy()
    i2 = two.plan(early_stop=True).cpu().numpy()
    assert two._parts is not None and len(two._parts) == 3 and one._parts is None
    assert np.array_equal(i1, i2, equal_nan=True)
    _assert_same(one, two)
    assert int((two.active == 0).sum().item()) > 0, "the workload should let some scenes terminate"


def test_small_batches_and_bare_iterate_are_not_pipelined(dev):
    make, _ = _engines(dev, 4, 64)
    eng = make()
    eng.iterate(0)
    eng.plan(early_stop=False)
    assert eng._parts is None


def test_pipelined_bench_workload_matches_oracle(dev):
    """bench.py's configuration run the way bench.py runs it (three parts since round 5) against the oracle on scenes of every part."""
    from oracle.check import engine_vs_oracle
    from omg_planner_amd.engine import ChompEngine
    make, batch = _engines(dev, 100, 64, grid=64)
    eng = make()
    eng.pipeline = ChompEngine.layout(100, 64)["pipeline"]
    assert eng.pipeline == 3
    for phase in range(2):
        r = engine_vs_oracle(eng, batch, [0, 32, 33, 49, 66, 99], steps=3, pin_window=True)
        assert r["goal_idx_equal"], r
        assert r["max_traj_err"] <= 1e-6 and r["max_cost_rel_err"] <= 1e-5, r


def test_prepared_iteration_calls_check_their_tensors(dev):
    """ops.IterationCalls validates once what goalset_cost_layer / goal_update_optimize validate per call."""
    from omg_planner_amd import _lib, ops
    make, _ = _engines(dev, 3, 8)
    e = make()

    def build(**over):
        a = dict(robot=e.robot, P=e.P, scenes=e.scenes, goals=e.cv_goals, dt=e.cfg.time_interval, traj=e.traj, layer_out=(e.pot, e.pgrad, e.col),
                 goal_out=(e.goal_cost, e.goal_col), goal_set=e.goal_set, reach=e.reach, state=e.learner_state, goal_idx=e.goal_idx, start=e.start,
                 end=e.end, goal_rows=e.goal_rows, goal_point=e.goal_point, step_out=(e.grad, e.cost_traj, e.info), cost_vector=e.cost_vec,
                 active=e._active, goal_count=e.goal_count, eta=e.eta_s, scene_flags=e._scene_flags)
        a.update(over)
        return ops.IterationCalls(**a)

    build()
    with pytest.raises(_lib.OmgHipError):
        build(traj=e.traj.float())
    with pytest.raises(_lib.OmgHipError):
        build(traj=e.traj.cpu())
    with pytest.raises(_lib.OmgHipError):
        build(layer_out=(e.pot[:, :-1], e.pgrad, e.col))
    with pytest.raises(_lib.OmgHipError):
        build(goal_idx=e.goal_idx.long())
    with pytest.raises(_lib.OmgHipError):
        build(active=e._active[:-1])
    # the engine builds new calls when one of its tensors is rebound, and the result stays that of the general path
    ref = make()
    for t in range(5):
        e.iterate(t)
        ref.separate_launches = True
        ref.iterate(t)
    first = e._hot[1]
    e.traj = e.traj.clone()
    e.iterate(5)
    ref.iterate(5)
    assert e._hot[1] is not first
    _assert_same(e, ref)


@pytest.mark.parametrize("S,early", [(3, True), (48, True), (48, False)])
def test_captured_plan_replays_to_the_same_bits(dev, S, early):
    """ChompEngine.capture_plan: the whole plan as one HIP graph (with 48 x 64 items also its two-stream pipeline) against
    plan(), twice from the same fresh state and once after the state was disturbed in between."""
    make, _ = _engines(dev, S, 64)
    ref = make()
    ref.plan(early_stop=early)
    eng = make()
    fresh = eng.snapshot()
    graph = eng.capture_plan(early_stop=early)
    assert (eng.t, eng.step_count) == (0, 0) and np.array_equal(eng.traj.cpu().numpy(), fresh["traj"].cpu().numpy())
    for rep in range(2):
        eng.restore(fresh)
        if rep == 1:
            eng.iterate(0)        # something else happened on the engine in between
            eng.restore(fresh)
        info = graph.replay()
        torch.cuda.synchronize()
        assert info is eng.info
        for k in ("traj", "info", "goal_idx", "learner_state", "end", "goal_rows"):
            assert np.array_equal(getattr(eng, k).cpu().numpy(), getattr(ref, k).cpu().numpy(), equal_nan=True), (rep, k)
        assert np.array_equal(eng.active.cpu().numpy(), ref.active.cpu().numpy())


def test_whole_plan_at_bench_size_follows_the_oracle(dev):
    """All 70 iterations of a plan (50 goal-selecting + 20 smoothing, shrinking goal-set window, weight schedule) on bench.py's 100
    scenes with the pipelined engine, two of the scenes followed by the oracle-driven loop from the same state: the bar is
    north_star's 1e-4 on trajectory states and costs, with the same goal choices all the way."""
    from oracle.check import engine_vs_oracle
    make, batch = _engines(dev, 100, 64, grid=64)
    eng = make()
    eng.pipeline = 2
    eng.select_initial_goal()
    r = engine_vs_oracle(eng, batch, [3, 96], steps=eng.cfg.optim_steps + eng.cfg.extra_smooth_steps, pin_window=False)
    assert r["goal_idx_equal"], r
    assert r["max_traj_err"] <= 1e-4 and r["max_cost_rel_err"] <= 1e-4, r


@pytest.mark.parametrize("S,early,alg", [(48, True, "MD"), (3, False, "MD"), (6, True, "Proj")])
def test_fixed_goal_iterations_through_the_prepared_calls_leave_the_same_bits(dev, S, early, alg):
    """The plan's last cfg.extra_smooth_steps iterations (goal fixed: layer launch + step) run through ops.IterationCalls
    (ChompEngine._iterate_hot_fixed) instead of the general path's checked calls: same entry points, same tensors, same bits —
    pipelined and not, with and without the early stop, and for a rule that never selects a goal."""
    import bench
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(S, 64, 30, 32, 0, False)
    out = []
    try:
        for hot in (True, False):
            ChompEngine.HOT_FIXED_GOAL = hot
            e = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=alg)
            e.plan(early_stop=early)
            torch.cuda.synchronize()
            out.append(e)
    finally:
        ChompEngine.HOT_FIXED_GOAL = True
    _assert_same(out[0], out[1])
