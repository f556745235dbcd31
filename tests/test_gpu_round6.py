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


# ------------------------------------------------------------------------------------------------
# omgx_plan_persistent: K iterations of every scene in one launch == K x (omgx_goalset_cost_layer + omgx_goal_update_optimize)
# ------------------------------------------------------------------------------------------------
_CMP = ("traj", "info", "learner_state", "goal_idx", "grad", "cost_traj", "pot", "pgrad", "col", "goal_cost", "goal_col", "end", "goal_rows",
        "goal_point", "cost_vec", "end_pose", "wp_pose")


def _pair(dev, S, G, n, alg, objects=4, grid=24, standoff=False, goal_counts=None, cfg_over=None):
    import bench
    from omg_planner_amd import scenes as sc
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(S, G, n, grid, 3, False, num_objects=objects)
    for k, v in (cfg_over or {}).items():
        setattr(cfg, k, v)
    reach = None
    if standoff:
        cfg.use_standoff = True
        c = cfg.reach_tail_length
        reach = np.stack([[np.concatenate([sc.linear_init(g - np.array([0.1, -0.05, 0.1, 0.15, 0, -0.1, 0.1, 0, 0]), g, c - 1), g[None]], 0) for g in goals[s]]
                          for s in range(S)])
    mk = lambda: ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, reach_grasps=reach, device=dev, ol_alg=alg, goal_counts=goal_counts)
    return mk(), mk()


def _same(a, b, what=""):
    torch.cuda.synchronize()
    for k in _CMP:
        x, y = getattr(a, k), getattr(b, k)
        assert torch.equal(x, y), (what, k, float((x.double() - y.double()).abs().max()))
    assert a.t == b.t and a.step_count == b.step_count


@pytest.mark.parametrize("alg,S,G,max_wg", [("MD", 6, 16, 0), ("FTL", 3, 9, 0), ("Exp", 9, 5, 24), ("FTC", 2, 33, 8), ("MD", 17, 12, 40)])
def test_persistent_launch_equals_the_iterations_it_replaces(dev, alg, S, G, max_wg):
    """run_persistent(range(K)) against K calls of iterate(): every tensor the iterations leave, bit for bit — with all of the
    chip's slots and with a handful of workgroups (long queues, every scene migrating between XCDs)."""
    a, b = _pair(dev, S, G, 30, alg)
    for e in (a, b):
        e.select_initial_goal()
        e.pose_hand_over(True)
    K = 12
    for t in range(K):
        a.iterate(t)
    b.run_persistent(range(K), max_workgroups=max_wg)
    _same(a, b, "first launch")
    st = b.persistent_status()
    assert st["failure"] == 0 and st["scenes_finished"] == S == st["scenes_planned"] and st["activations"] == S * K, st
    # a second launch from where the first one ended (the queue is re-initialised by every call)
    for t in range(K, K + 5):
        a.iterate(t)
    b.run_persistent(range(K, K + 5), max_workgroups=max_wg)
    _same(a, b, "second launch")


def test_persistent_launch_fixed_goal_iterations_and_early_stop(dev):
    """A whole plan's iteration sequence: cfg.optim_steps goal-selecting iterations, then fixed-goal ones (layer + step only), with
    early stop (planner.py:626: a scene that terminates leaves the loop; active[s] <- 0)."""
    a, b = _pair(dev, 7, 8, 30, "MD", cfg_over={"optim_steps": 9, "extra_smooth_steps": 6})
    for e in (a, b):
        e.select_initial_goal()
        e.pose_hand_over(True)
    ts = list(range(15))
    for t in ts:
        a.iterate(t, early_stop=True)
    b.run_persistent(ts, early_stop=True)
    _same(a, b)
    assert torch.equal(a.active, b.active)


def test_persistent_launch_ragged_goal_sets_and_standoff(dev):
    a, b = _pair(dev, 5, 12, 30, "MD", standoff=True, goal_counts=[12, 3, 7, 1, 12])
    for e in (a, b):
        e.select_initial_goal()
        e.pose_hand_over(True)
    for t in range(8):
        a.iterate(t)
    b.run_persistent(range(8))
    _same(a, b)


def test_persistent_launch_fifty_waypoints_a_dozen_objects(dev):
    """BASELINE config 5's shape: 50 waypoints (a goal workgroup's LDS no longer admits five per CU), 12 obstacles + table."""
    a, b = _pair(dev, 3, 16, 50, "MD", objects=12, grid=32)
    for e in (a, b):
        e.select_initial_goal()
        e.pose_hand_over(True)
    for t in range(6):
        a.iterate(t)
    b.run_persistent(range(6))
    _same(a, b)


def test_persistent_launch_whole_plan_against_the_oracle(dev):
    """Twenty planner iterations, each a persistent launch, against the oracle-driven loop: the persistent path on its own, not only
    against the launches it replaces."""
    from omg_planner_amd.engine import ChompEngine
    from oracle.check import engine_vs_oracle
    import bench
    cfg, model, batch, start, goals = bench.build_workload(3, 16, 30, 24, 11, False)

    class Persistent(ChompEngine):
        def iterate(self, t, early_stop=False):
            self.run_persistent([t], early_stop)
    eng = Persistent(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    eng.select_initial_goal()
    out = engine_vs_oracle(eng, batch, [0, 1, 2], steps=20, pin_window=False)
    assert out["ok"] and out["max_traj_err"] < 1e-6, out


@pytest.mark.parametrize("update_cus", [1, 3])
def test_persistent_launch_with_dedicated_update_cus(dev, update_cus):
    """A launch that fills the chip, with the scenes' updates served by the workgroups of dedicated CUs (omg_persist.h: roles): the
    same bits as the launched iterations — who runs an update changes nothing."""
    a, b = _pair(dev, 24, 64, 30, "MD", grid=32)
    for e in (a, b):
        e.select_initial_goal()
        e.pose_hand_over(True)
    for t in range(8):
        a.iterate(t)
    b.run_persistent(range(8), update_cus=update_cus)
    _same(a, b)
    st = b.persistent_status()
    assert st["failure"] == 0 and st["scenes_finished"] == 24, st


@pytest.mark.parametrize("G", [128, 200])
def test_persistent_launch_many_goals(dev, G):
    """More than 64 goals: the persistent launch's learner holds two (G <= 128) or four goals per lane — other instantiations of the
    same float64 code than the update kernels', whose fused multiply-adds the compiler may place differently: the learner's state agrees
    to the last bits (1e-15), everything downstream of the chosen goal bit for bit."""
    a, b = _pair(dev, 3, G, 30, "MD", grid=24)
    for e in (a, b):
        e.select_initial_goal()
        e.pose_hand_over(True)
    for t in range(6):
        a.iterate(t)
    b.run_persistent(range(6))
    torch.cuda.synchronize()
    for k in _CMP:
        x, y = getattr(a, k), getattr(b, k)
        if k == "learner_state":
            np.testing.assert_allclose(y.cpu().numpy(), x.cpu().numpy(), rtol=1e-12, atol=1e-15)
        else:
            assert torch.equal(x, y), (k, float((x.double() - y.double()).abs().max()))


def _set_wide(on, max8=-1, max6=-1, long6=-1):
    import ctypes as C
    from omg_planner_amd import _lib
    f = _lib.lib().omgx_debug_set_wide
    f.argtypes = [C.c_int, C.c_longlong, C.c_longlong, C.c_longlong]
    f.restype = None
    f(on, max8, max6, long6)


@pytest.mark.parametrize("waves,alg,S,G,n,kw", [
    (8, "MD", 5, 24, 30, {}), (6, "MD", 5, 24, 30, {}), (8, "FTL", 3, 70, 30, {"standoff": True}), (6, "Exp", 9, 16, 17, {}),
    (8, "MD", 4, 12, 30, {"goal_counts": [12, 5, 1, 9]}), (6, "MD", 2, 40, 32, {"objects": 9}), (8, "FTC", 3, 8, 5, {}),
    (6, "MD", 3, 10, 50, {"objects": 12}), (6, "FTL", 2, 7, 64, {}), (6, "Exp", 2, 9, 41, {"standoff": True})])
def test_wide_workgroups_change_no_bit(dev, waves, alg, S, G, n, kw):
    """The WIDE instantiations of the batch kernel (six / eight waves per goal workgroup, chosen for launches small enough to be resident
    at once; omg_kernels.hip gs_wide_waves): masks, tiles and the exact sum are those of four waves — every tensor the iterations
    leave is bit for bit the same, through the dispatch-schedule measurement and the scheduled launches too."""
    try:
        _set_wide(0)
        a, b = _pair(dev, S, G, n, alg, **kw)
        for e in (a, b):
            e.select_initial_goal()
            e.pose_hand_over(True)
        for t in range(8):
            a.iterate(t)
        torch.cuda.synchronize()
        _set_wide(1, 1 << 40 if waves == 8 else 0, 1 << 40 if waves == 6 else 0, 1 << 40)
        for t in range(8):
            b.iterate(t)
        _same(a, b, f"{waves} waves")
        # the stand-alone goal-set entry point (Cost.batch_obstacle_cost's numbers) as well
        ca, cb = a.goal_cost.clone(), b.goal_cost.clone()
        assert torch.equal(ca, cb)
    finally:
        _set_wide(1, -2, -2, -2)  # back to the built-in rule


@pytest.mark.parametrize("S,G,n,alg,proj,update_cus", [(1, 1, 12, "FTL", False, 1), (1, 5, 20, "Exp", False, 1), (1, 2, 41, "FTL", True, 2), (2, 2, 12, "MD", True, 2),
                                                      (3, 1, 12, "FTL", False, 8), (1, 8, 30, "MD", True, 2)])
def test_persistent_launch_of_a_tiny_plan_ignores_dedicated_update_cus(dev, S, G, n, alg, proj, update_cus):
    """Fuzz campaign r06final (seed 12011, 7 of 6000): a launch of a handful of workgroups with `update_cus` > 0 made every workgroup an
    update workgroup — the first CUs of an XCD to report ARE all its CUs then — and nobody ran an item (2 s of waiting, failure code 5,
    tensors never written).  Dedicated update CUs exist only in launches that fill the chip; and a failure inside a launch raises at the
    engine's next join()."""
    import bench
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(S, G, n, 20, 3, False, num_objects=3)
    cfg.goal_set_proj = proj
    mk = lambda: ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=alg)
    a, b = mk(), mk()
    for e in (a, b):
        e.select_initial_goal()
        e.pose_hand_over(True)
    for t0 in (0, 1, 2, 4):
        ts = [t0] if t0 < 2 else [t0, t0 + 1]
        for t in ts:
            a.iterate(t)
        b.run_persistent(ts, update_cus=update_cus)
        b.join()  # raises on a failure code
        _same(a, b, f"iterations {ts}")
        st = b.persistent_status()
        assert st["failure"] == 0 and st["scenes_finished"] == S, st


def test_the_library_picks_the_goal_workgroups_width_by_the_window(dev, request):
    """omg_kernels.hip gs_wide_waves / launch_goalset: four waves per goal workgroup, EIGHT in plans of 57 .. 64 waypoints (the launch's LDS
    follows the trajectory layer's n configurations: two workgroups per CU either way), split goals and latency mode never wide, waypoint ranges only behind their hook — read
    back through the library's test probe after real launches of the engine."""
    import ctypes as C
    import bench
    from omg_planner_amd import _lib
    from omg_planner_amd.engine import ChompEngine
    probe = _lib.lib().omgx_debug_last_goalset_variant
    probe.restype = C.c_int

    def variant(S, G, n, **kw):
        cfg, model, batch, start, goals = bench.build_workload(S, G, n, 16, 1, False)
        e = ChompEngine(model, batch, cfg, start, goals, device=dev, ol_alg="MD", **kw)
        e.iterate(0)
        torch.cuda.synchronize()
        return probe()

    assert variant(3, 8, 30) == 4 and variant(3, 8, 50) == 4 and variant(2, 8, 56) == 4
    assert variant(2, 8, 64) == 8 and variant(2, 8, 60) == 8 and variant(2, 8, 57) == 8
    assert variant(2, 8, 64, goal_parts=2) == 0x200 | 4 and variant(2, 8, 50, goal_parts=2) == 0x200 | 4
    assert variant(1, 8, 64, latency_mode=True) & 0x400
    f = _lib.lib().omgx_debug_set_range
    f.argtypes, f.restype = [C.c_int], None
    try:
        f(40)
        assert variant(2, 8, 50, goal_parts=2) == 0x300 | 4 and variant(2, 8, 30, goal_parts=2) == 0x200 | 4
    finally:
        f(-1)


def test_an_engine_laid_out_for_plans_agrees_with_the_default_one(dev):
    """ChompEngine.auto(for_plan=True): two to four scenes plan in latency mode (round 6: whole plans are faster there, the pinned step is
    not).  Same goals, trajectories at 1e-9 (a goal's cost is summed in another number of parts), the oracle's bars."""
    import bench
    from omg_planner_amd.engine import ChompEngine
    from oracle.check import engine_vs_oracle
    cfg, model, batch, start, goals = bench.build_workload(3, 16, 30, 24, 5, False)
    a = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    b = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", for_plan=True)
    assert b.latency and not a.latency and b.layout_used["latency_mode"]
    a.plan(early_stop=False)
    b.plan(early_stop=False)
    torch.cuda.synchronize()
    assert torch.equal(a.goal_idx, b.goal_idx)
    np.testing.assert_allclose(b.traj.cpu().numpy(), a.traj.cpu().numpy(), rtol=0, atol=1e-9)
    c = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD", for_plan=True)
    c.select_initial_goal()
    r = engine_vs_oracle(c, batch, [0, 1, 2], steps=12, pin_window=False)
    assert r["goal_idx_equal"] and r["max_traj_err"] <= 1e-6 and r["max_cost_rel_err"] <= 1e-5, r
