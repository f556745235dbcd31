"""Kinematics pre-pass (ABI 10: k_goalset_kin -> k_goalset_queue<..., PRE>; ChompEngine(prepass=True)).  The goals' link poses and
row masks are computed by a launch of their own (one lane per (goal, configuration)) into a workspace in HBM and the goal
workgroups start from there.  What has to hold: EVERY output bit of every goal-set entry point equals the single-launch form's —
goal costs, collision counts, layer outputs, handed-over poses — in the batch layout, with split goals, in latency mode, with
ragged goal sets, `active` masks, dispatch schedules and at every window length; whole plans (pipelined, as a graph) leave the
same bits behind; and the oracle agrees as before."""
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


def _make(dev, S, G, counts=None, grid=32, alg="MD", n=30, **kw):
    import bench
    from omg_planner_amd.engine import ChompEngine
    cfg, model, batch, start, goals = bench.build_workload(S, G, n, grid, 0, False)
    return ChompEngine(model, batch, copy.deepcopy(cfg), start, goals, device=dev, ol_alg=alg, goal_counts=counts, **kw), batch


def _nan_like(*ts):
    return tuple(torch.full_like(t, float("nan")) for t in ts)


def _eq(a, b):
    return np.array_equal(a.cpu().numpy(), b.cpu().numpy(), equal_nan=True)


@pytest.mark.parametrize("n_rem", [30, 29, 17, 8, 5, 2, 1])
@pytest.mark.parametrize("goal_parts", [1, 2, 4])
def test_prepass_launch_writes_the_single_launch_bits(dev, n_rem, goal_parts):
    from omg_planner_amd import ops
    S, G = 5, 24
    counts = np.array([24, 11, 17, 1, 24])
    eng, _ = _make(dev, S, G, counts)
    active = torch.tensor([1, 1, 0, 1, 1], dtype=torch.int32, device=dev)
    ts = eng.traj[:, 30 - n_rem]
    NP = ops.goalset_parts(n_rem, goal_parts) if goal_parts > 1 else 1
    outs = []
    for pre in (False, True):
        lay = _nan_like(eng.pot, eng.pgrad, eng.col)
        pc = torch.full((S, G * NP), float("nan"), dtype=torch.float32, device=dev)
        pl = torch.full_like(pc, float("nan"))
        poses = torch.full((S, 30, 10, 12), float("nan"), dtype=torch.float64, device=dev)
        ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay, out=(pc, pl),
                               goal_count=eng.goal_count, active=active, goal_parts=goal_parts, layer_poses=poses, prepass=pre)
        # ... and under a dispatch schedule built from the measuring launch
        work = torch.zeros(S * G * NP, dtype=torch.int32, device=dev)
        sched = ops.goalset_schedule(None, S, G, goal_count=eng.goal_count, parts=NP, device=dev)
        pc2, pl2 = _nan_like(pc, pl)
        ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay, out=(pc2, pl2),
                               goal_count=eng.goal_count, active=active, goal_parts=goal_parts, schedule=sched, work=work, prepass=pre)
        sched2 = ops.goalset_schedule(work, S, G, goal_count=eng.goal_count, parts=NP, active=active)
        pc3, pl3 = _nan_like(pc, pl)
        ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay, out=(pc3, pl3),
                               goal_count=eng.goal_count, active=active, goal_parts=goal_parts, schedule=sched2, prepass=pre)
        torch.cuda.synchronize()
        assert _eq(pc, pc2) and _eq(pl, pl2)
        # (an active-aware schedule leaves the inactive scene's items out: what it does write equals the plain launch)
        m = ~torch.isnan(pc3)
        assert torch.equal(pc3[m], pc[m]) and bool(m.any())
        outs.append((pc, pl, poses, *lay))
    for a, b in zip(*outs):
        assert _eq(a, b)
    c = outs[0][0].reshape(S, G, NP)
    assert torch.isnan(c[2]).all() and not torch.isnan(c[0]).any() and torch.isnan(c[1, 11:]).all() and not torch.isnan(c[3, 0]).any()


@pytest.mark.parametrize("n_rem", [30, 13, 3])
def test_prepass_in_latency_mode_and_cost_only_entry_points(dev, n_rem):
    from omg_planner_amd import ops
    S, G = 2, 20
    eng, batch = _make(dev, S, G, np.array([20, 7]))
    ts = eng.traj[:, 30 - n_rem]
    NP = ops.goalset_parts(n_rem, 4)
    outs = []
    for pre in (False, True):
        lay = _nan_like(eng.pot, eng.pgrad, eng.col)
        pc = torch.full((S, G * NP), float("nan"), dtype=torch.float32, device=dev)
        pl = torch.full_like(pc, float("nan"))
        ops.goalset_cost_layer_tiled(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, eng.traj, lay, (pc, pl),
                                     goal_count=eng.goal_count, goal_parts=4, layer_link_groups=10, layer_config_block=4, spread=True, prepass=pre)
        c1, l1, _ = ops.goalset_cost(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, n_rem, eng.cfg.time_interval, goal_count=eng.goal_count,
                                     out=_nan_like(pc[:, :G].contiguous(), pc[:, :G].contiguous()), prepass=pre)
        torch.cuda.synchronize()
        outs.append((pc, pl, c1, l1, *lay))
    for a, b in zip(*outs):
        assert _eq(a, b)
    from oracle import oracle as orc
    gc, _ = orc.goalset_cost(eng.model.blob(), eng.P, batch.subset(0, 1), ts[0:1].cpu().numpy(), eng.cv_goals[0:1].cpu().numpy(), n_rem,
                             eng.cfg.time_interval)
    np.testing.assert_allclose(outs[1][2][0].cpu().numpy(), np.asarray(gc).reshape(-1), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n,G", [(50, 9), (64, 5), (7, 33)])
def test_prepass_at_other_window_lengths(dev, n, G):
    """50 waypoints (BASELINE config 5), 64 (the ABI's maximum: 65 configurations, one goal per wave in two passes), 7 (eight goals
    per wave)."""
    from omg_planner_amd import ops
    S = 3
    eng, _ = _make(dev, S, G, n=n)
    outs = []
    for pre in (False, True):
        lay = _nan_like(eng.pot, eng.pgrad, eng.col)
        c, l = ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, eng.traj[:, 0], eng.cv_goals, n, eng.cfg.time_interval, eng.traj, lay, prepass=pre)
        torch.cuda.synchronize()
        outs.append((c, l, *lay))
    for a, b in zip(*outs):
        assert _eq(a, b)


@pytest.mark.parametrize("kw,alg", [(dict(), "MD"), (dict(goal_parts=2), "FTL"), (dict(latency_mode=True), "MD")])
def test_a_plan_with_the_prepass_leaves_the_same_bits(dev, kw, alg):
    """plan() — pipelined where the layout pipelines, early stop, ragged goal sets — and the same plan replayed as one HIP graph."""
    S, G = (2, 16) if kw.get("latency_mode") else (7, 16)
    counts = np.array([16, 9, 12, 16, 5, 16, 3][:S])
    out = []
    for pre in (False, True):
        e, _ = _make(dev, S, G, counts, alg=alg, prepass=pre, **kw)
        e.plan(early_stop=True)
        torch.cuda.synchronize()
        out.append({k: getattr(e, k).cpu().numpy().copy() for k in ("traj", "info", "goal_idx", "learner_state", "grad", "cost_traj", "end", "goal_rows", "pot", "col", "goal_cost")})
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k], equal_nan=True), k
    c, _ = _make(dev, S, G, counts, alg=alg, prepass=True, **kw)
    fresh = c.snapshot()
    graph = c.capture_plan(early_stop=True)
    c.restore(fresh)
    graph.replay()
    torch.cuda.synchronize()
    for k in ("traj", "info", "goal_idx", "learner_state", "end", "goal_rows"):
        assert np.array_equal(getattr(c, k).cpu().numpy(), out[0][k], equal_nan=True), k


def test_prepass_engine_follows_the_oracle(dev):
    from oracle.check import engine_vs_oracle
    d, batch = _make(dev, 2, 16, prepass=True)
    d.select_initial_goal()
    r = engine_vs_oracle(d, batch, [0, 1], steps=12, pin_window=False)
    assert r["goal_idx_equal"] and r["max_traj_err"] <= 1e-6 and r["max_cost_rel_err"] <= 1e-5, r


def test_prepass_workspace_is_checked(dev):
    from omg_planner_amd import _lib, ops
    eng, _ = _make(dev, 2, 8)
    lay = _nan_like(eng.pot, eng.pgrad, eng.col)
    small = torch.empty(64, dtype=torch.uint8, device=dev)
    with pytest.raises(_lib.OmgHipError):
        ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, eng.traj[:, 0], eng.cv_goals, 30, eng.cfg.time_interval, eng.traj, lay, prepass=small)
    need = _lib.lib().omgx_goalset_workspace_bytes(2, 8, 30, eng.P)
    assert need == ((2 * 8 * (90 * 31 * 8 + 300 * 4) + 15) // 16) * 16
    own = torch.empty(need, dtype=torch.uint8, device=dev)
    a = ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, eng.traj[:, 0], eng.cv_goals, 30, eng.cfg.time_interval, eng.traj, lay, prepass=own)
    b = ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, eng.traj[:, 0], eng.cv_goals, 30, eng.cfg.time_interval, eng.traj, lay)
    torch.cuda.synchronize()
    assert _eq(a[0], b[0]) and _eq(a[1], b[1])
    # the poses in the workspace are the kinematics of the interpolated configurations: [goal][link][component][configuration]
    w = own[: 2 * 8 * 90 * 31 * 8].view(torch.float64).reshape(2, 8, 10, 9, 31).cpu().numpy()
    lin = np.linspace(0.0, 1.0, 32)[1:-1]
    q0, qg = eng.traj[0, 0].cpu().numpy(), eng.cv_goals[0, 3].cpu().numpy()
    cfgs = np.concatenate([q0[None], q0[None] + lin[:, None] * (qg - q0)[None]])
    T = ops.pose_table(eng.robot, eng.P, torch.as_tensor(cfgs, device=dev)).cpu().numpy().reshape(31, 10, 12)
    np.testing.assert_allclose(w[0, 3, :, 0:6, :].transpose(2, 0, 1), T[:, :, 0:6], rtol=0, atol=1e-12)
    np.testing.assert_allclose(w[0, 3, :, 6:9, :].transpose(2, 0, 1), T[:, :, 9:12], rtol=0, atol=1e-12)


def test_huge_goal_costs_stay_within_the_oracles_bar(dev):
    """The exact (order-independent) goal sum holds while a goal's cost stays below 2^17 (omg_goalset_queue.h: tsum).  A scene whose
    volumes hold potentials of thousands pushes the sums far beyond: the batch kernel, the split launch and the oracle must still
    agree at the tolerances of every other test (the additions then round at 2^-53 relative)."""
    import bench
    from omg_planner_amd import ops, scenes as sc
    from omg_planner_amd.engine import ChompEngine
    from oracle import oracle as orc
    cfg, model, batch, start, goals = bench.build_workload(2, 12, 30, 32, 0, False)
    pool = np.array(batch.pool, np.float32)
    pool[pool < 0.05] -= 4000.0  # inside / near every object: potentials of ~4e3, goal costs of ~1e6-1e7
    big = sc.SceneBatch(batch.objects, batch.scene_begin, pool)
    eng = ChompEngine(model, big, copy.deepcopy(cfg), start, goals, device=dev, ol_alg="MD")
    ts = eng.traj[:, 0]
    lay = _nan_like(eng.pot, eng.pgrad, eng.col)
    c1, _ = ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, 30, eng.cfg.time_interval, eng.traj, lay)
    pc = torch.zeros((2, 12 * 4), dtype=torch.float32, device=dev)
    ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, 30, eng.cfg.time_interval, eng.traj, lay, out=(pc, torch.zeros_like(pc)), goal_parts=4)
    torch.cuda.synchronize()
    tot = pc.reshape(2, 12, 4).sum(-1).cpu().numpy()
    a = c1.cpu().numpy()
    assert a.max() > 2.0 ** 17
    np.testing.assert_allclose(tot, a, rtol=2e-6)
    for s in range(2):
        gc, _ = orc.goalset_cost(eng.model.blob(), eng.P, big.subset(s, s + 1), ts[s:s + 1].cpu().numpy(), eng.cv_goals[s:s + 1].cpu().numpy(), 30, eng.cfg.time_interval)
        np.testing.assert_allclose(a[s], np.asarray(gc).reshape(-1), rtol=1e-5)


@pytest.mark.parametrize("goal_parts", [1, 2])
def test_goal_costs_do_not_depend_on_who_draws_which_tile(dev, goal_parts):
    """Since round 5 the waves of a goal workgroup DRAW their tiles from an LDS counter: which wave computes which pair depends
    on timing.  The goal's cost is an exact sum and the counts are integers, so a launch
    repeated under different loads — alone, beside a second stream that keeps the chip busy — must leave the same bits, and they
    must be the bits of a goal split over several workgroups' draws added up by the learner's rule (one float32 rounding per
    part: compared here part by part between repetitions only)."""
    from omg_planner_amd import ops
    S, G = 24, 64
    eng, _ = _make(dev, S, G, grid=32)
    NP = ops.goalset_parts(30, goal_parts) if goal_parts > 1 else 1
    ts = eng.traj[:, 0]

    def launch():
        lay = _nan_like(eng.pot, eng.pgrad, eng.col)
        pc = torch.full((S, G * NP), float("nan"), dtype=torch.float32, device=dev)
        pl = torch.full_like(pc, float("nan"))
        ops.goalset_cost_layer(eng.robot, eng.P, eng.scenes, ts, eng.cv_goals, 30, eng.cfg.time_interval, eng.traj, lay, out=(pc, pl),
                               goal_parts=goal_parts)
        return (pc, pl) + lay

    first = launch()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    noise_a = torch.randn(4096, 4096, device=dev)
    for rep in range(4):
        if rep % 2:  # something else on the chip while the launch runs: other waves win other draws
            with torch.cuda.stream(side):
                for _ in range(3):
                    noise_a = torch.tanh(noise_a @ noise_a * 1e-3)
        again = launch()
        torch.cuda.synchronize()
        for x, y in zip(first, again):
            assert _eq(x, y), rep
    assert torch.isfinite(first[0]).all() and float(first[0].abs().sum()) > 0.0
