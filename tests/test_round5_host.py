"""Host logic added in round 5 (no GPU): the links' bounding balls (robot blob BALL), the promises of the drop-in planner loop,
the workload matching of the roofline inputs."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_bounding_ball_is_the_smallest_ball_and_sits_in_the_blob():
    from omg_planner_amd import robot as rb
    rng = np.random.RandomState(3)
    for trial in range(60):
        pts = rng.normal(size=(rng.randint(1, 17), 3)) * rng.uniform(0.01, 0.5) + rng.normal(size=3)
        c, r = rb.bounding_ball(pts)[:3], rb.bounding_ball(pts)[3]
        d = np.linalg.norm(pts - c, axis=1)
        assert d.max() <= r <= d.max() * (1 + 1e-9) + 1e-12          # holds every point, radius = the farthest one
        on = int((d > r * (1 - 1e-7) - 1e-12).sum())
        assert on >= min(2, len(pts))                                 # a smallest ball is held by at least two points
        # no centre nearby does better (a smallest enclosing ball is the unique minimiser)
        for _ in range(20):
            c2 = c + rng.normal(size=3) * 0.02 * max(r, 1e-3)
            assert np.linalg.norm(pts - c2, axis=1).max() >= r * (1 - 1e-9)
    m = rb.PandaModel(seed=0)
    b, P = m.blob(), m.points_per_link
    D = 528 + 30 * P
    assert b.size == D + 356 + 30 * P
    pts = b[D + 246: D + 246 + 30 * P].reshape(10, P, 3)
    rad = b[D + 306 + 30 * P: D + 316 + 30 * P]
    ball = b[D + 316 + 30 * P:].reshape(10, 4)
    for l in range(10):
        assert np.linalg.norm(pts[l] - ball[l, :3], axis=1).max() <= ball[l, 3]
        assert ball[l, 3] <= rad[l] + 1e-12                           # never larger than the ball about the frame origin
    assert (ball[:, 3] < 0.8 * rad).sum() >= 5                        # ... and much smaller for most links: what the culling gains


class _FakeLoop:
    def __init__(self, value):
        self.value, self.forced = value, 0
        self.traj_obj = type("T", (), {"goal_set": np.arange(45.0).reshape(5, 9)})()

    def force(self, promise):
        self.forced += 1
        promise._value = self.value


def test_promises_of_the_drop_in_loop_behave_like_their_values():
    from omg_planner_amd.device_loop import LazyBool, LazyIndex, _LazyEnd
    loop = _FakeLoop(3)
    i = LazyIndex(loop)
    assert not i.resolved() and loop.forced == 0
    sel = [i]                       # planner.py:617 stores it without looking
    assert loop.forced == 0
    assert int(i) == 3 and loop.forced == 1 and i.resolved()
    assert i == 3 and not (i != 3) and i < 4 and i >= 3 and hash(i) == hash(3) and f"{i}" == "3" and repr(sel) == "[3]"
    assert np.arange(10)[i] == 3 and [10, 11, 12, 13][i] == 13 and i + 1 == 4 and 5 - i == 2
    assert int(np.asarray(i)) == 3 and loop.forced == 1           # resolved once
    b = LazyBool(_FakeLoop(True))
    assert bool(b) is True and repr(b) == "True"
    end = _LazyEnd(loop, i)
    assert end.shape == (9,) and len(end) == 9
    np.testing.assert_array_equal(np.asarray(end), loop.traj_obj.goal_set[3])
    np.testing.assert_array_equal(end - 1.0, loop.traj_obj.goal_set[3] - 1.0)
    assert end[2] == loop.traj_obj.goal_set[3][2] and float(end.sum()) == float(loop.traj_obj.goal_set[3].sum())


def test_roofline_inputs_match_by_shape_and_scale_by_scenes(tmp_path):
    from tools.roofline import match_inputs, roofline_block
    base = {"valu_wave_insts_per_launch": 30.0e6, "valu_issue_cycles_per_launch": 75.0e6, "hbm_bytes_per_launch": 6.0e7, "l2_hit_rate": 0.9,
            "from_profiles_tag": "x", "calibration_tag": "c", "useful": {"exact_path_valu_share": 0.3, "pairs": {"contributing": 9}}}
    wl = {"scenes": 100, "goals": 64, "waypoints": 30, "points_per_link": 15, "grid": 64, "pipeline": 2, "objects": 5}
    other = dict(base, from_profiles_tag="y", valu_wave_insts_per_launch=6.0e6, valu_issue_cycles_per_launch=15.0e6,
                 workload={"scenes": 13, "goals": 128, "waypoints": 30, "points_per_link": 15, "grid": 64, "pipeline": 3, "objects": 5})
    inp = dict(base, workload=wl, others=[other])
    e, sc = match_inputs(inp, wl)
    assert e["from_profiles_tag"] == "x" and sc == 1.0
    e, sc = match_inputs(inp, dict(wl, scenes=50, pipeline=1))            # the same 50 scenes per launch
    assert e["from_profiles_tag"] == "x" and sc == 1.0
    e, sc = match_inputs(inp, dict(wl, pipeline=3))                        # three parts of 33.3 scenes
    assert e["from_profiles_tag"] == "x" and sc == pytest.approx((100 / 3) / 50)
    e, sc = match_inputs(inp, dict(other["workload"], scenes=12))          # the 12-scene shard of the 13-scene profile
    assert e["from_profiles_tag"] == "y" and sc == pytest.approx(12 / 13)
    assert match_inputs(inp, dict(wl, waypoints=50)) == (None, None) and match_inputs(inp, dict(wl, objects=13)) == (None, None)
    old = {k: v for k, v in wl.items() if k != "objects"}                   # profiles from before round 5 do not name the objects: 5
    assert match_inputs(dict(base, workload=old), wl)[1] == 1.0
    f = tmp_path / "inputs.json"
    f.write_text(json.dumps(inp))
    r = roofline_block(f, 0.15, 40, 5, 1.0e9, dict(wl, pipeline=3), launches_per_step=3, ms_per_step=0.19)
    assert r["counts_scaled"] == pytest.approx((100 / 3) / 50) and r["valu_wave_insts_per_launch"] == pytest.approx(30.0e6 * (100 / 3) / 50)
    assert r["from_profiles_tag"] == "x" and r["profiled_workload"] == wl and 0 < r["frac"] < 1 and r["useful_frac"] == 0.3
    # the tracked file: one primary workload, the other shapes bench.py times beside it, no shape left without counts
    tracked = json.loads((ROOT / "profiles" / "roofline_inputs.json").read_text())
    shapes = [tracked["workload"]] + [e["workload"] for e in tracked["others"]]
    assert {(w["scenes"], w["goals"], w["waypoints"], w["objects"]) for w in shapes} >= {(100, 64, 30, 5), (13, 128, 30, 5), (16, 64, 50, 13), (100, 128, 30, 5)}
    for w in shapes:
        assert match_inputs(tracked, w)[1] == 1.0
