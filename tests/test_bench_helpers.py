"""CPU test of bench.py's host-side pieces: workload construction and the cpu_baseline leg (the oracle timed on a
bounded sample); the timed GPU region itself needs an MI355X."""
import numpy as np


def test_workload_and_cpu_baseline():
    import bench
    cfg, model, batch, start, goals = bench.build_workload(3, 4, 30, 16, seed0=0, share_grids=False)
    assert batch.num_scenes == 3 and goals.shape == (3, 4, 9) and start.shape == (3, 9)
    assert len(batch.objects) == 15 and batch.objects["grid_offset"].max() < batch.pool.size
    # private volumes: no two records share a grid
    assert len(set(batch.objects["grid_offset"].tolist())) == 15
    shared = bench.build_workload(3, 4, 30, 16, seed0=0, share_grids=True)[2]
    assert shared.pool.size < batch.pool.size
    # goals are grasp-like: hand within ~0.2 m of the target
    from omg_planner_amd import scenes as sc
    pos, _ = sc.hand_pose(model, goals[0])
    assert np.linalg.norm(pos - sc.make_tabletop_scene(0, grid=16).objects[0].pose_mat[:3, 3], axis=1).max() < 0.45
    out = bench.cpu_baseline(cfg, model, batch, start, goals, 30, budget_s=0.5)
    assert out["kind"] == "port" and out["unit"] == "iterations/s" and out["value"] > 0 and out["cores"] >= 1


def test_roofline_block_per_launch_and_chip_level(tmp_path):
    """tools/roofline.py: one launch per step -> counts / launch duration; k launches in flight at once (pipelined engine) ->
    counts of a step's launches / the step's wall time, the per-launch figure beside it; the peak is the chip's SIMD cycles over
    the mean issue cycles of the kernel's OWN instruction mix (frac = issue cycles occupied / SIMD cycles passed); counts are
    dropped when the tracked inputs belong to another workload; the committed inputs reproduce the committed bench line."""
    import json
    import pytest
    from tools.roofline import CLOCK_GHZ, ROOT, SIMDS, roofline_block
    wl = {"scenes": 100, "goals": 64, "pipeline": 2}
    inp = tmp_path / "inputs.json"
    inp.write_text(json.dumps({"from_profiles_tag": "t", "calibration_tag": "c", "workload": wl, "valu_wave_insts_per_launch": 60.0e6,
                               "valu_issue_cycles_per_launch": 150.0e6, "hbm_bytes_per_launch": 1.0e8, "l2_hit_rate": 0.95,
                               "useful": {"exact_path_valu_share": 0.3, "pairs": {"tested": 100, "box_survivors": 15, "contributing": 9}}}))
    one = roofline_block(inp, 0.25, 40, 5, 1.0e10, wl)
    assert one["basis"].startswith("per launch") and one["achieved"] == pytest.approx(60.0e6 / 0.25e-3 / 1e9)
    assert one["peak"] == pytest.approx(SIMDS * CLOCK_GHZ / 2.5) and one["mean_issue_cycles_per_instr"] == pytest.approx(2.5)
    assert one["frac"] == pytest.approx(150.0e6 / (SIMDS * CLOCK_GHZ * 1e9 * 0.25e-3)) and one["per_launch"]["achieved"] == pytest.approx(one["achieved"])
    assert one["useful_frac"] == 0.3 and one["pairs"]["contributing"] == 9
    two = roofline_block(inp, 0.25, 40, 5, 1.0e10, wl, launches_per_step=2, ms_per_step=0.30)
    assert two["basis"].startswith("chip level") and two["achieved"] == pytest.approx(2 * 60.0e6 / 0.30e-3 / 1e9)
    assert two["frac"] == pytest.approx(2 * 150.0e6 / (SIMDS * CLOCK_GHZ * 1e9 * 0.30e-3))
    assert two["per_launch"]["achieved"] == pytest.approx(60.0e6 / 0.25e-3 / 1e9) and two["frac"] <= 1.0
    assert two["hbm_real"]["GBs"] == pytest.approx(2 * 1.0e8 / 0.30e-3 / 1e9)
    with pytest.raises(ValueError):
        roofline_block(inp, 0.25, 40, 5, 1.0e10, wl, launches_per_step=2)
    other = roofline_block(inp, 0.25, 40, 5, 1.0e10, dict(wl, goals=128))
    assert other["achieved"] is None and other["frac"] is None and other["from_profiles_tag"] is None and other["pairs"] is None
    # the tracked inputs and the bench line they belong to
    tracked = json.loads((ROOT / "profiles" / "roofline_inputs.json").read_text())
    bench = ROOT / "profiles" / f"{tracked['from_profiles_tag']}_bench.json"
    line = json.loads([l for l in bench.read_text().splitlines() if l.startswith("{")][-1])
    r = line["roofline"]
    again = roofline_block(ROOT / "profiles" / "roofline_inputs.json", r["avg_launch_ms"], r["launches"], r["timing_stride"],
                           r["algorithmic_bytes_per_launch"], tracked["workload"], r.get("launches_per_step", 1), r.get("ms_per_step"))
    assert again["frac"] == pytest.approx(r["frac"], rel=1e-6) and 0.0 < again["frac"] <= 1.0
    assert again["useful_frac"] == r["useful_frac"] and again["pairs"] == r["pairs"]


def test_issue_price_list_comes_from_the_residency_checked_calibration():
    """tools/roofline.calibration(): only rows whose waves were co-resident (overlap >= 0.95, W waves on every SIMD) with >= 4
    waves per SIMD count; the committed table prices a full-rate instruction at 2 cycles and f64 / conversions at 4."""
    from tools.roofline import calibration, issue_cycles, MIX_CLASSES
    cal = calibration()
    assert 1.95 < cal["v_fma_f32"] < 2.15 and 1.95 < cal["v_add_u32"] < 2.15
    assert 3.9 < cal["v_fma_f64"] < 4.1 and 3.9 < cal["v_cvt_f32_f64"] < 4.1 and 3.9 < cal["v_mad_u32_u24"] < 4.1
    assert 7.5 < cal["v_rcp_f32"] < 8.5
    mix = {c: 1.0e6 for c in MIX_CLASSES}
    cycles, parts = issue_cycles(mix, 20.0e6, cal)
    assert parts["OTHER"]["count"] == 20.0e6 - len(MIX_CLASSES) * 1.0e6
    assert cycles == sum(p["count"] * p["cycles_each"] for p in parts.values()) and 2.0 * 20.0e6 < cycles < 4.0 * 20.0e6
