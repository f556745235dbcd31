#!/usr/bin/env python3
"""Roofline numbers of the dominant kernel (k_goalset_queue<2>) — ONE formula, two users.

    python tools/roofline.py --tag r02h                     # profiles/r02h_*.csv -> profiles/roofline_inputs.json, prints the block
    python tools/roofline.py --tag r02h --bench BENCH.json  # recompute the `roofline` object of a bench.py line and compare

bench.py calls roofline_block() with the launch duration it measured live (HIP events attached to the dispatch); the
per-launch COUNTS it divides come from profiles/roofline_inputs.json, which this script derives from the committed rocprofv3
summaries of the same command (tools/collect_profiles.sh <tag>) and which carries that tag — so every number of the
block is either measured in the run (avg_launch_ms) or traceable to profiles/<tag>_pmc_*.csv (`from_profiles_tag`).

What bounds the kernel: VALU instruction ISSUE.  A wave64 VALU instruction occupies its SIMD for 4 cycles whatever its
type (f32, f64, conversions alike — tools/valu_rates.hip: 4.1-5.0 cycles per instruction per SIMD at 8 waves per SIMD; SQ_ACTIVE_INST_VALU
counts exactly one quad-cycle per instruction), so the chip retires at most 1024 SIMDs x 2.4 GHz / 4 = 614.4 G
wave-instructions/s.  `achieved` = the launch's SQ_INSTS_VALU / its duration.  HBM and MFMA are not the bound: the launch
moves ~0.2 GB (2-3 % of what 8 TB/s would carry in its duration) and has no matrix-shaped work.
"""
from __future__ import annotations

import argparse
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4.0   # G wave-instructions/s, see the module docstring
KERNEL = "k_goalset_queue<2, false>"


def _pmc(tag: str, name: str, counter: str, profiles: Path):
    f = profiles / f"{tag}_pmc_{name}.csv"
    if not f.exists():
        return None
    for r in csv.DictReader(open(f)):
        if KERNEL in r["kernel"] and r["counter"] == counter:
            return float(r["mean_per_dispatch"])
    return None


def derive_inputs(tag: str, profiles: Path = ROOT / "profiles") -> dict:
    """Per-launch counts of the dominant kernel from profiles/<tag>_*.csv."""
    valu = _pmc(tag, "SQ_INSTS", "SQ_INSTS_VALU", profiles)
    fetch, write = _pmc(tag, "FETCH_SIZE", "FETCH_SIZE", profiles), _pmc(tag, "WRITE_SIZE", "WRITE_SIZE", profiles)
    hit, miss = _pmc(tag, "TCC", "TCC_HIT_sum", profiles), _pmc(tag, "TCC", "TCC_MISS_sum", profiles)
    wave_cyc, wait = _pmc(tag, "SQ_WAIT", "SQ_WAVE_CYCLES", profiles), _pmc(tag, "SQ_WAIT", "SQ_WAIT_ANY", profiles)
    gui = _pmc(tag, "GRBM", "GRBM_GUI_ACTIVE", profiles)
    stats_ns = None
    f = profiles / f"{tag}_kernel_stats.csv"
    if f.exists():
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Name"]:
                stats_ns = float(r["AverageNs"])
    cfg_file = profiles / f"{tag}_workload.json"
    out = {
        "from_profiles_tag": tag,
        "kernel": "k_goalset_queue<2, false> (goal-set batch + trajectory layer: kinematics, culling, SDF lookups, arc-length cost)",
        "workload": json.loads(cfg_file.read_text()) if cfg_file.exists() else None,
        "valu_wave_insts_per_launch": valu,
        # rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the
        # bytes of wide coalesced reads -> doubled as prescribed (an upper bound for this gather pattern); WRITE_SIZE as reported
        "hbm_bytes_per_launch": None if fetch is None else (2.0 * fetch + (write or 0.0)) * 1024.0,
        "fetch_size_kib": fetch, "write_size_kib": write,
        "l2_hit_rate": None if not (hit and miss) else hit / (hit + miss),
        "wave_wait_share": None if not (wave_cyc and wait) else wait / wave_cyc,
        "avg_launch_ns_rocprof_kernel_trace": stats_ns,
        "gpu_clock_ghz_during_pmc": None if not (gui and stats_ns) else gui / 8.0 / stats_ns,
    }
    return out


def roofline_block(inputs_file, avg_launch_ms: float, launches: int, timing_stride: int, algorithmic_bytes: float, workload: dict,
                   launches_per_step: int = 1, ms_per_step: float | None = None) -> dict:
    """The `roofline` object of bench.py's JSON line.

    One launch of the kernel per step (launches_per_step = 1): achieved = counts per launch / its average duration (HIP events).
    Pipelined engine (k launches per step, in flight at the same time on k streams): a launch's duration then covers time in
    which it shares the machine with its sibling, so the per-launch rate (still reported, `per_launch`) counts the machine k
    times; achieved = counts of the k launches of a step / the step's wall time — the share of the chip's issue rate that
    the kernel's instructions take over the whole timed region, update launches and gaps included."""
    inp = json.loads(Path(inputs_file).read_text()) if Path(inputs_file).exists() else {}
    same = inp.get("workload") is not None and all(inp["workload"].get(k) == v for k, v in workload.items())
    valu = inp.get("valu_wave_insts_per_launch") if same else None
    hbm = inp.get("hbm_bytes_per_launch") if same else None
    sec = avg_launch_ms * 1e-3
    per_launch = None if valu is None else valu / sec / 1e9
    chip = launches_per_step > 1
    if chip and ms_per_step is None:
        raise ValueError("ms_per_step is needed when several launches of the kernel are in flight at once")
    sec_eff = ms_per_step * 1e-3 / launches_per_step if chip else sec  # wall time per launch of the kernel
    achieved = None if valu is None else valu / sec_eff / 1e9
    block = {
        "bound": "valu-issue",
        "kernel": inp.get("kernel", KERNEL),
        "achieved": achieved, "peak": VALU_PEAK_GINST, "unit": "G wave-instr/s",
        "frac": None if achieved is None else achieved / VALU_PEAK_GINST,
        "basis": (f"chip level: counts of the {launches_per_step} launches of a step / ms_per_step ({launches_per_step} launches in flight at once)"
                  if chip else "per launch: counts of a launch / avg_launch_ms"),
        "per_launch": {"achieved": per_launch, "frac": None if per_launch is None else per_launch / VALU_PEAK_GINST},
        "launches_per_step": launches_per_step, "ms_per_step": ms_per_step,
        "avg_launch_ms": avg_launch_ms, "launches": launches, "timing_stride": timing_stride,
        "valu_wave_insts_per_launch": valu,
        "traffic": hbm,
        "hbm_real": None if hbm is None else {"bytes_per_launch": hbm, "GBs": hbm / sec_eff / 1e9, "frac_of_peak": hbm / sec_eff / 1e9 / HBM_PEAK_GBS,
                                               "peak_GBs": HBM_PEAK_GBS, "l2_hit_rate": inp.get("l2_hit_rate")},
        "algorithmic_equiv_GBs": algorithmic_bytes / sec_eff / 1e9,
        "algorithmic_bytes_per_launch": algorithmic_bytes,
        "from_profiles_tag": inp.get("from_profiles_tag") if same else None,
        "note": "bound = VALU instruction issue (1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction); counts per launch from "
                "profiles/<from_profiles_tag>_pmc_*.csv via tools/roofline.py, durations measured in this run; algorithmic_equiv_GBs "
                "(SURVEY 8d: 32 + 128 O bytes per point) is not a fraction of anything: the kernel retires 85 % of the pairs in "
                "registers" + ("" if same else "; profiles/roofline_inputs.json is for another workload: counts omitted"),
    }
    return block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--bench", default=None, help="a bench.py JSON line to recompute (default: profiles/<tag>_bench.json)")
    ap.add_argument("--no-write", action="store_true")
    args = ap.parse_args()
    prof = ROOT / "profiles"
    inp = derive_inputs(args.tag, prof)
    import tempfile
    inputs_path = prof / "roofline_inputs.json"
    if args.no_write:  # recompute an older tag's bench line from that tag's own CSVs without touching the tracked inputs
        inputs_path = Path(tempfile.mkdtemp()) / "roofline_inputs.json"
    inputs_path.write_text(json.dumps(inp, indent=1) + "\n")
    print(json.dumps(inp, indent=1))
    bench = Path(args.bench) if args.bench else prof / f"{args.tag}_bench.json"
    if bench.exists():
        line = [l for l in bench.read_text().splitlines() if l.startswith("{")][-1]
        b = json.loads(line)
        r = b["roofline"]
        mine = roofline_block(inputs_path, r["avg_launch_ms"], r["launches"], r["timing_stride"], r["algorithmic_bytes_per_launch"],
                              inp["workload"] or {}, r.get("launches_per_step", 1), r.get("ms_per_step"))
        ok = True
        for k in ("achieved", "frac", "traffic", "algorithmic_equiv_GBs"):
            a_, b_ = mine[k], r.get(k)
            same = (a_ is None and b_ is None) or (a_ is not None and b_ is not None and abs(a_ - b_) <= 5e-4 * max(abs(a_), abs(b_)))
            ok &= same
            print(f"{k:24s} recomputed {a_!r:24} bench line {b_!r:24} {'ok' if same else 'DIFFERENT'}")
        if inp["avg_launch_ns_rocprof_kernel_trace"]:
            print(f"avg launch: bench.py HIP events {r['avg_launch_ms'] * 1e3:.1f} us, rocprofv3 --kernel-trace --stats {inp['avg_launch_ns_rocprof_kernel_trace'] / 1e3:.1f} us")
        print("roofline frac", mine["frac"], "<= 1:", mine["frac"] is not None and mine["frac"] <= 1.0)
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
