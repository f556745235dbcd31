"""omgx_goalset_schedule (include/omg_hip.h section 7) against a numpy restatement of its integer specification, and the
properties the goal-set launch relies on: every kept (scene, goal) item appears exactly once, an XCD's column holds a
contiguous run of the scene-major list, the pieces carry equal work (as measured; clamped when a piece would not fit its slots
otherwise), and the result of the launch does not depend on the order.

The schedule is dispatch policy of THIS implementation (the reference has no counterpart: omg/online_learner.py:128-148
evaluates the goal set as one batch); what the reference fixes is the launch's result, which test_gpu_parity.py checks under
the schedule.
"""
from __future__ import annotations

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU: torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def schedule_mirror(work, S, G, active=None, goal_count=None, slack=2):
    """The specification in include/omg_hip.h, with Python integers."""
    n_slots = (slack * S * G + 7) // 8 + 2
    sched = np.full(n_slots * 8, -1, np.int64)
    kept = np.ones((S, G), bool)
    if active is not None:
        kept &= (np.asarray(active) != 0)[:, None]
    if goal_count is not None:
        kept &= np.arange(G)[None, :] < np.asarray(goal_count)[:, None]
    w = np.ones((S, G), np.int64) if work is None else np.maximum(np.asarray(work, np.int64).reshape(S, G), 1)
    n_all = max(int(kept.sum()), 1)
    total = int(w[kept].sum())
    lo = max((10 * total) // (14 * n_all), 1)
    hi = slack * lo
    wc = np.clip(w, lo, hi)
    Ws = [int(w[s][kept[s]].sum()) for s in range(S)]
    Wc = [int(wc[s][kept[s]].sum()) for s in range(S)]
    order = sorted(range(S), key=lambda s: (-Ws[s], s))
    total_c, total_r = max(sum(Wc), 1), max(total, 1)
    pos, cum, cum_r, items = 0, 0, 0, []
    for s in order:
        goals = sorted((g for g in range(G) if kept[s, g]), key=lambda g: (-int(w[s, g]), g))
        for g in goals:
            x = min(7, (8 * (2 * cum + int(wc[s, g]))) // (2 * total_c))          # the list cut by clamped work
            xr = min(7, (8 * (2 * cum_r + int(w[s, g]))) // (2 * total_r))        # ... by the work as measured
            items.append((pos, x, xr, s * G + g))
            pos += 1
            cum += int(wc[s, g])
            cum_r += int(w[s, g])
    # the raw cut when every piece fits its slots, the clamped one (whose pieces always do) otherwise
    counts = np.bincount([xr for _, _, xr, _ in items], minlength=8) if items else np.zeros(8, int)
    raw = bool((counts <= n_slots).all())
    first = {}
    for p, x, xr, _ in items:
        first.setdefault(xr if raw else x, p)
    for p, x, xr, it in items:
        xx = xr if raw else x
        sched[(p - first[xx]) * 8 + xx] = it
    schedule_mirror.last_cut_was_raw = raw
    return sched, (w if raw else wc), kept


CASES = [
    # S, G, ragged, masked, work
    (100, 64, False, False, "measured"),
    (100, 64, False, False, "narrow"),   # scene weights within a factor of two, like measured durations: the raw cut fits
    (33, 64, True, True, "narrow"),
    (100, 64, False, True, "measured"),
    (13, 64, True, False, "measured"),
    (12, 64, False, False, "uniform"),
    (1, 64, False, False, "measured"),
    (3, 5, True, True, "ties"),
    (256, 7, True, True, "measured"),
    (40, 100, False, False, "skewed"),
    (200, 100, True, True, "measured"),   # 20 000 items: beyond the LDS-staged path
    (300, 40, False, False, "measured"),  # 12 000 items but more scenes than the staged path takes
]


@pytest.mark.parametrize("S,G,ragged,masked,kind", CASES)
def test_schedule_matches_the_integer_specification(dev, S, G, ragged, masked, kind):
    from omg_planner_amd import ops
    rng = np.random.RandomState(S * 131 + G)
    if kind == "uniform":
        work = None
    elif kind == "ties":
        work = rng.randint(0, 3, S * G).astype(np.int32)  # zeros count as 1; many equal weights: index order decides
    elif kind == "skewed":
        work = (rng.pareto(1.2, S * G) * 2000 + 100).astype(np.int32)  # heavy tail: the clamp band is what bounds a piece
    elif kind == "narrow":
        work = (rng.randint(20000, 36000, S)[:, None] * rng.uniform(0.6, 1.6, (S, G))).astype(np.int32).ravel()
    else:
        base = rng.randint(4000, 30000, S)[:, None]
        work = (base * rng.uniform(0.3, 1.5, (S, G))).astype(np.int32).ravel()
    goal_count = rng.randint(1, G + 1, S).astype(np.int32) if ragged else None
    active = (rng.uniform(size=S) < 0.6).astype(np.int32) if masked else None
    if masked and S > 1:
        active[0], active[-1] = 1, 0
    want, wc, kept = schedule_mirror(work, S, G, active, goal_count)
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    got = ops.goalset_schedule(t(work), S, G, active=t(active), goal_count=t(goal_count), device=dev).cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_array_equal(got, want)
    # every kept item exactly once, nothing else
    items = got[got >= 0]
    assert sorted(items.tolist()) == np.flatnonzero(kept.ravel()).tolist()
    # column x is filled from row 0 without holes, and the pieces carry equal clamped work up to one item
    cols = got.reshape(-1, 8)
    share = wc[kept].sum() / 8.0
    for x in range(8):
        col = cols[:, x]
        k = int((col >= 0).sum())
        assert (col[:k] >= 0).all() and (col[k:] < 0).all()
        if k:
            load = wc.ravel()[col[:k]].sum()
            assert abs(load - share) <= wc[kept].max(), (x, load, share)


def test_both_cuts_are_exercised(dev):
    """The measured-work cut is the usual one; a heavy-tailed weight distribution overflows a piece's slots and falls back to the
    clamped cut.  Both must occur among the cases above (the device result is compared with the mirror there)."""
    rng = np.random.RandomState(40 * 131 + 100)
    skew = (rng.pareto(1.2, 40 * 100) * 2000 + 100).astype(np.int32)
    schedule_mirror(skew, 40, 100)
    assert schedule_mirror.last_cut_was_raw is False
    rng = np.random.RandomState(100 * 131 + 64)
    meas = (rng.randint(20000, 36000, 100)[:, None] * rng.uniform(0.6, 1.6, (100, 64))).astype(np.int32).ravel()
    schedule_mirror(meas, 100, 64)
    assert schedule_mirror.last_cut_was_raw is True


@pytest.mark.parametrize("S,G,ragged,masked,kind,parts", [(13, 128, False, False, "measured", 1), (5, 64, True, True, "ties", 1),
                                                           (25, 64, True, False, "skewed", 1), (4, 64, False, False, "measured", 4),
                                                           (1, 64, False, False, "measured", 1)])
def test_longest_first_inside_an_xcd(dev, S, G, ragged, masked, kind, parts):
    """omgx_goalset_schedule_ordered(OMGX_SCHEDULE_LONGEST_FIRST): every item on the XCD the scene-major list gives it, an XCD's items
    by decreasing work (ties: lower item first); above OMGX_SCHEDULE_LONGEST_FIRST_MAX_ITEMS items the scene-major order itself."""
    from omg_planner_amd import _lib, ops
    rng = np.random.RandomState(S * 17 + G + parts)
    Gi = G * parts
    if kind == "ties":
        work = rng.randint(0, 3, S * Gi).astype(np.int32)
    elif kind == "skewed":
        work = (rng.pareto(1.2, S * Gi) * 2000 + 100).astype(np.int32)
    else:
        work = (rng.randint(4000, 30000, S)[:, None] * rng.uniform(0.3, 1.5, (S, Gi))).astype(np.int32).ravel()
    goal_count = rng.randint(1, G + 1, S).astype(np.int32) if ragged else None
    active = (rng.uniform(size=S) < 0.6).astype(np.int32) if masked else None
    if masked:
        active[0] = 1
    base, _, kept = schedule_mirror(work, S, Gi, active, None if goal_count is None else goal_count * parts)
    w = np.maximum(work.astype(np.int64), 1)
    want = np.full_like(base, -1)
    for x in range(8):
        col = base.reshape(-1, 8)[:, x]
        items = col[col >= 0]
        items = np.array(sorted(items.tolist(), key=lambda i: (-int(w[i]), i)), np.int64)
        want.reshape(-1, 8)[: len(items), x] = items
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    got = ops.goalset_schedule(t(work), S, G, active=t(active), goal_count=t(goal_count), device=dev, parts=parts, longest_first=True).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert sorted(got[got >= 0].tolist()) == np.flatnonzero(kept.ravel()).tolist()


def test_longest_first_falls_back_to_scene_major_for_large_launches(dev):
    from omg_planner_amd import _lib, ops
    S, G = 160, 64  # 10 240 items > OMGX_SCHEDULE_LONGEST_FIRST_MAX_ITEMS
    assert S * G > _lib.SCHEDULE_LONGEST_FIRST_MAX_ITEMS
    work = torch.from_numpy(np.random.RandomState(3).randint(1000, 30000, S * G).astype(np.int32)).to(dev)
    a = ops.goalset_schedule(work, S, G, device=dev).cpu().numpy()
    b = ops.goalset_schedule(work, S, G, device=dev, longest_first=True).cpu().numpy()
    np.testing.assert_array_equal(a, b)


def test_schedule_rejects_what_it_cannot_hold(dev):
    from omg_planner_amd import _lib, ops
    with pytest.raises(_lib.OmgHipError):
        ops.goalset_schedule(None, 2048, 64, active=torch.ones(2048, dtype=torch.int32, device=dev))  # > 65536 items
    with pytest.raises(_lib.OmgHipError):
        ops.goalset_schedule(torch.ones(10, dtype=torch.int32, device=dev), 4, 4)  # work shorter than S*G


def test_engine_reschedules_without_terminated_scenes(dev):
    """Early stop: the schedule of the next goal-set launch holds exactly the goals of the scenes still running, and the plan
    ends with the same trajectories as without any schedule."""
    import bench
    from omg_planner_amd.engine import ChompEngine
    S, G = 40, 64
    cfg, model, batch, start, goals = bench.build_workload(S, G, 30, 32, 0, False)
    out = {}
    for mode in ("scheduled", "plain"):
        eng = ChompEngine(model, batch, cfg, start, goals, device=dev, ol_alg="MD")
        eng.auto_schedule, eng.reschedule_every = mode == "scheduled", 1
        eng.plan(early_stop=True)
        torch.cuda.synchronize()
        if mode == "scheduled" and eng._masked and eng._measured:
            live = eng._active.cpu().numpy() != 0
            items = eng.schedule.cpu().numpy()
            items = items[items >= 0]
            # the schedule was built from the mask of the LAST launch: all its scenes were running then
            assert len(items) % G == 0 and len(set((items // G).tolist())) * G == len(items)
            assert set(np.flatnonzero(live).tolist()) <= set((items // G).tolist())
        out[mode] = (eng.traj.cpu().numpy().copy(), eng.goal_idx.cpu().numpy().copy())
    np.testing.assert_array_equal(out["scheduled"][1], out["plain"][1])
    np.testing.assert_array_equal(out["scheduled"][0], out["plain"][0])
