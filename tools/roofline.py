#!/usr/bin/env python3
"""Roofline numbers of the dominant kernel (k_goalset_queue<2>) — ONE formula, two users.

    python tools/roofline.py --tag r03x                     # profiles/r03x_*.csv -> profiles/roofline_inputs.json, prints the block
    python tools/roofline.py --tag r03x --bench BENCH.json  # recompute the `roofline` object of a bench.py line and compare

bench.py calls roofline_block() with the launch duration it measured live (HIP events attached to the dispatch); the
per-launch COUNTS it divides come from profiles/roofline_inputs.json, which this script derives from the committed rocprofv3
summaries of the same command (tools/collect_profiles.sh <tag>) and which carries that tag — so every number of the
block is either measured in the run (avg_launch_ms) or traceable to profiles/<tag>_*.csv (`from_profiles_tag`).

What bounds the kernel: VALU instruction ISSUE — it moves ~0.1 GB per launch (a tenth of what 8 TB/s would carry in its
duration) and has no matrix-shaped work.  The price of an instruction is NOT uniform (round 2 assumed 4 cycles for all;
VERDICT r02 item 1).  tools/valu_peak.hip (profiles/r03d_valu_peak.{csv,txt}, cross-checked by GRBM_GUI_ACTIVE / SQ_INSTS_VALU in
profiles/r03d_valu_peak_pmc.csv, both within 2 %) measures, with >= 4 waves per SIMD resident for a 3 ms window:

    2 cycles  per wave64 instruction per SIMD   v_fma_f32 / v_add_f32 / v_mul_f32 / v_and_b32 / v_add_u32 (full rate, as
                                                MI355X_MICROARCH.md says: SIMD-32, 64 lanes in 2 cycles)
    4 cycles                                    every f64 op, v_cvt_*, v_mad_u32_u24, v_lshl_add_u32, DPP moves, v_pk_fma_f32
    3 cycles  per instruction                   v_cmp + v_cndmask pairs (6 per pair)
    8 cycles                                    v_rcp_f32 / v_sqrt_f32
    one wave alone issues an instruction every 5.8-10 cycles whatever its kind: a SIMD needs >= 4 ready waves to reach these rates

so the issue time a launch needs at least is  sum over categories  count x cycles  / (1024 SIMDs x clock), with the counts of
rocprofv3's SQ_INSTS_VALU_* counters (profiles/<tag>_pmc_MIX{1,2}.csv).  Instructions those counters leave uncategorised (moves,
compares, selects, bit operations, lane operations: ~40 %) and the int32 class are priced at the full rate (2 cycles), int64 and
conversions at 4: a LOWER bound of the issue time, i.e. the reported `frac` is if anything too LOW.
`frac` = that issue time / the wall time the launches took (at the data-sheet clock of 2.4 GHz; the CUs sustain 2.0-2.2 GHz
under this kind of load, see the GHz column of the calibration).  `useful_frac` says how much of the issued work is the exact
path (the lookups that can contribute), the rest being triage (kinematics, culling, far tests, queueing).
"""
from __future__ import annotations

import argparse
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
SIMDS = 256 * 4
CLOCK_GHZ = 2.4           # data-sheet maximum
KERNEL = "k_goalset_queue<2, false, false, false, false, 4>"  # <LB, STAMP, LAT, SPLIT, PRE, W>: no work stamps, batch layout, whole goals, own kinematics, four waves
KERNEL_WIDE = "k_goalset_queue<2, false, false, false, false, 8>"  # ... on eight waves (windows of 57-64 waypoints)
KERNEL_R5 = "k_goalset_queue<2, false, false, false, false>"  # the same instantiation before the sixth template parameter (round 6's wide workgroups): profiles r05*, r06a*
KERNEL_R4 = "k_goalset_queue<2, false, false, false>"      # the same instantiation in round 4 (four template parameters): profiles r04*
KERNEL_OLD = "k_goalset_queue<2, false, false>"            # ... and before round 4 (three): profiles up to r03h


KERNEL_SPLIT = "k_goalset_queue<2, false, false, true, false"  # the batch kernel with a goal's tiles over two or more workgroups (long windows, small batches); with or without W


def is_dominant(name: str) -> bool:
    return any(k in name for k in (KERNEL, KERNEL_WIDE, KERNEL_R5, KERNEL_SPLIT, KERNEL_R4, KERNEL_OLD))
CALIBRATION_TAG = "r03d"  # profiles/<tag>_valu_peak.csv

# SQ_INSTS_VALU_* class -> the calibration row that prices it (cheapest member of the class: lower bound of the issue time)
MIX_CLASSES = {
    "ADD_F32": "v_add_f32", "MUL_F32": "v_mul_f32", "FMA_F32": "v_fma_f32", "TRANS_F32": "v_rcp_f32", "CVT": "v_cvt_f32_f64",
    "INT32": "v_add_u32", "INT64": "v_mad_u32_u24", "ADD_F64": "v_add_f64", "MUL_F64": "v_mul_f64", "FMA_F64": "v_fma_f64",
    "TRANS_F64": "v_rcp_f32",  # no f64 transcendental in the calibration: at least the f32 one's price
}
OTHER_ROW = "v_and_b32"  # uncategorised instructions: priced at the full rate


def calibration(tag: str = CALIBRATION_TAG, profiles: Path = ROOT / "profiles") -> dict:
    """instruction -> cycles per wave64 instruction per SIMD: the lowest figure among the residency-checked rows
    (overlap >= 0.95, every SIMD with exactly W waves) with >= 3 waves per SIMD (full-rate instructions reach their 2 cycles
    from 4-5 waves on, the 8-cycle transcendentals already at 3)."""
    out: dict = {}
    f = profiles / f"{tag}_valu_peak.csv"
    if not f.exists():
        return out
    for r in csv.DictReader(open(f)):
        if int(r["waves_per_simd"]) < 3 or float(r["overlap"]) < 0.95 or int(r["simds_with_other_wave_count"]) != 0:
            continue
        c = float(r["cycles_per_instr_per_simd"])
        out[r["instruction"]] = min(out.get(r["instruction"], c), c)
    return out


def _pmc(tag: str, name: str, counter: str, profiles: Path):
    f = profiles / f"{tag}_pmc_{name}.csv"
    if not f.exists():
        return None
    for r in csv.DictReader(open(f)):
        if is_dominant(r["kernel"]) and r["counter"] == counter:
            return float(r["mean_per_dispatch"])
    return None


def issue_cycles(mix: dict, total: float, cal: dict):
    """(lower bound of the) SIMD cycles the launch's VALU instructions occupy, summed over all SIMDs; per-class breakdown."""
    if not mix or not cal or OTHER_ROW not in cal:
        return None, None
    parts = {}
    seen = 0.0
    for cls, row in MIX_CLASSES.items():
        n = mix.get(cls)
        if n is None or row not in cal:
            return None, None
        parts[cls] = {"count": n, "cycles_each": cal[row], "priced_as": row}
        seen += n
    parts["OTHER"] = {"count": max(total - seen, 0.0), "cycles_each": cal[OTHER_ROW], "priced_as": OTHER_ROW}
    return sum(p["count"] * p["cycles_each"] for p in parts.values()), parts


def derive_inputs(tag: str, profiles: Path = ROOT / "profiles") -> dict:
    """Per-launch counts of the dominant kernel from profiles/<tag>_*.csv."""
    valu = _pmc(tag, "SQ_INSTS", "SQ_INSTS_VALU", profiles) or _pmc(tag, "MIX1", "SQ_INSTS_VALU", profiles)
    fetch, write = _pmc(tag, "FETCH_SIZE", "FETCH_SIZE", profiles), _pmc(tag, "WRITE_SIZE", "WRITE_SIZE", profiles)
    hit, miss = _pmc(tag, "TCC", "TCC_HIT_sum", profiles), _pmc(tag, "TCC", "TCC_MISS_sum", profiles)
    wave_cyc, wait = _pmc(tag, "SQ_WAIT", "SQ_WAVE_CYCLES", profiles), _pmc(tag, "SQ_WAIT", "SQ_WAIT_ANY", profiles)
    gui = _pmc(tag, "GRBM", "GRBM_GUI_ACTIVE", profiles)
    mix = {}
    for cls in MIX_CLASSES:
        v = _pmc(tag, "MIX1", f"SQ_INSTS_VALU_{cls}", profiles)
        if v is None:
            v = _pmc(tag, "MIX2", f"SQ_INSTS_VALU_{cls}", profiles)
        if v is not None:
            mix[cls] = v
    cal = calibration()
    cycles, parts = issue_cycles(mix if len(mix) == len(MIX_CLASSES) else None, valu or 0.0, cal)
    stats_ns = None
    kernel_name = None  # the instantiation that dominates THIS workload (whole goals on four / eight waves, split goals), as the trace names it
    f = profiles / f"{tag}_kernel_stats.csv"
    if f.exists():
        best = -1.0
        for r in csv.DictReader(open(f)):
            if is_dominant(r["Name"]) and float(r.get("TotalDurationNs") or r.get("AverageNs") or 0) > best:
                best = float(r.get("TotalDurationNs") or r.get("AverageNs") or 0)
                stats_ns = float(r["AverageNs"])
                kernel_name = r["Name"].replace("void ", "").split("(")[0]
    cfg_file = profiles / f"{tag}_workload.json"
    useful_file = profiles / f"{tag}_useful.json"  # tools/gs_block_counts.py (counting build) + the exact-path ablation pass
    useful = json.loads(useful_file.read_text()) if useful_file.exists() else None
    out = {
        "from_profiles_tag": tag,
        "calibration_tag": CALIBRATION_TAG,
        "kernel": (kernel_name or KERNEL) + " (goal-set batch + trajectory layer: kinematics, culling, SDF lookups, arc-length cost)",
        "workload": json.loads(cfg_file.read_text()) if cfg_file.exists() else None,
        "valu_wave_insts_per_launch": valu,
        "valu_issue_cycles_per_launch": cycles,   # summed over the SIMDs; a lower bound (module docstring)
        "valu_mix": parts,
        # rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the
        # bytes of wide coalesced reads -> doubled as prescribed (an upper bound for this gather pattern); WRITE_SIZE as reported
        "hbm_bytes_per_launch": None if fetch is None else (2.0 * fetch + (write or 0.0)) * 1024.0,
        "fetch_size_kib": fetch, "write_size_kib": write,
        "l2_hit_rate": None if not (hit and miss) else hit / (hit + miss),
        "wave_wait_share": None if not (wave_cyc and wait) else wait / wave_cyc,
        "avg_launch_ns_rocprof_kernel_trace": stats_ns,
        "gpu_clock_ghz_during_pmc": None if not (gui and stats_ns) else gui / 8.0 / stats_ns,
        "useful": useful,
    }
    return out


SHAPE_KEYS = ("goals", "waypoints", "points_per_link", "grid", "objects")  # what a goal workgroup's work depends on
DEFAULT_OBJECTS = 5  # profiles collected before round 5 do not name it: 4 obstacles + the table


def _entries(inp: dict):
    """All profiled workloads of an inputs file: the primary one (top level) and inp["others"]."""
    out = [inp] if inp.get("workload") else []
    return out + list(inp.get("others") or [])


def match_inputs(inp: dict, workload: dict):
    """-> (entry, scale) for the launch `workload` describes, or (None, None).  An entry fits when a goal workgroup does the same
    work in it — goals per scene, window, points per link, grid, objects per scene — and its per-launch counts are then scaled by
    the number of scenes a launch handles (scenes / pipeline parts): the counters are sums over workgroups, and a scene's
    workgroups do not know how many other scenes the launch holds.  scale == 1.0 for the profiled launch itself."""
    def shape(w):
        return tuple(w.get(k, DEFAULT_OBJECTS if k == "objects" else None) for k in SHAPE_KEYS)

    def per_launch(w):
        return float(w.get("scenes", 0)) / max(1, int(w.get("pipeline", 1)))
    want, n = shape(workload), per_launch(workload)
    best = None
    for e in _entries(inp):
        w = e.get("workload") or {}
        if shape(w) != want or per_launch(w) <= 0 or n <= 0:
            continue
        d = abs(per_launch(w) - n)
        if best is None or d < best[0]:
            best = (d, e, n / per_launch(w))
    return (None, None) if best is None else (best[1], best[2])


def roofline_block(inputs_file, avg_launch_ms: float, launches: int, timing_stride: int, algorithmic_bytes: float, workload: dict,
                   launches_per_step: int = 1, ms_per_step: float | None = None) -> dict:
    """The `roofline` object of bench.py's JSON line.

    One launch of the kernel per step (launches_per_step = 1): achieved = counts per launch / its average duration (HIP events).
    Pipelined engine (k launches per step, in flight at the same time on k streams): a launch's duration then covers time in
    which it shares the machine with its sibling, so the per-launch rate (still reported, `per_launch`) counts the machine k
    times; achieved = counts of the k launches of a step / the step's wall time — the share of the chip's issue rate that
    the kernel's instructions take over the whole timed region, update launches and gaps included.

    peak = 1024 SIMDs x 2.4 GHz / (mean issue cycles of THIS kernel's instructions), so that frac = achieved / peak =
    (issue cycles the instructions occupy) / (SIMD cycles that passed).

    The per-launch COUNTS come from the profiled workload of profiles/roofline_inputs.json that matches (match_inputs): the same
    work per goal workgroup, scaled to this launch's number of scenes (`counts_scaled`: 1.0 for the profiled launch itself)."""
    whole = json.loads(Path(inputs_file).read_text()) if Path(inputs_file).exists() else {}
    inp, scale = match_inputs(whole, workload)
    same = inp is not None
    inp = inp or {}
    k_ = (lambda v: None if v is None else v * scale) if same else (lambda v: None)
    valu = k_(inp.get("valu_wave_insts_per_launch"))
    cycles = k_(inp.get("valu_issue_cycles_per_launch"))
    hbm = k_(inp.get("hbm_bytes_per_launch"))
    useful = inp.get("useful") if same else None
    sec = avg_launch_ms * 1e-3
    chip = launches_per_step > 1
    if chip and ms_per_step is None:
        raise ValueError("ms_per_step is needed when several launches of the kernel are in flight at once")
    sec_eff = ms_per_step * 1e-3 / launches_per_step if chip else sec  # wall time per launch of the kernel
    mean_cycles = None if not (valu and cycles) else cycles / valu
    peak = None if mean_cycles is None else SIMDS * CLOCK_GHZ / mean_cycles
    achieved = None if valu is None else valu / sec_eff / 1e9
    per_launch = None if valu is None else valu / sec / 1e9
    frac = None if (achieved is None or peak is None) else achieved / peak
    block = {
        "bound": "valu-issue",
        "kernel": whole.get("kernel", KERNEL),
        "achieved": achieved, "peak": peak, "unit": "G wave-instr/s",
        "frac": frac,
        "mean_issue_cycles_per_instr": mean_cycles,
        "peak_if_every_instr_were_full_rate": SIMDS * CLOCK_GHZ / 2.0,
        "basis": (f"chip level: counts of the {launches_per_step} launches of a step / ms_per_step ({launches_per_step} launches in flight at once)"
                  if chip else "per launch: counts of a launch / avg_launch_ms"),
        "per_launch": {"achieved": per_launch, "frac": None if (per_launch is None or peak is None) else per_launch / peak},
        "launches_per_step": launches_per_step, "ms_per_step": ms_per_step,
        "avg_launch_ms": avg_launch_ms, "launches": launches, "timing_stride": timing_stride,
        "valu_wave_insts_per_launch": valu,
        "valu_issue_cycles_per_launch": cycles,
        # of everything issued, the share that is exact-path work (lookups of pairs that survived the triage), and what the
        # triage let through: pairs tested / surviving the row + box tests / actually contributing a potential or a collision
        "useful_frac": None if not useful else useful.get("exact_path_valu_share"),
        "pairs": None if not useful else useful.get("pairs"),
        "traffic": hbm,
        "hbm_real": None if hbm is None else {"bytes_per_launch": hbm, "GBs": hbm / sec_eff / 1e9, "frac_of_peak": hbm / sec_eff / 1e9 / HBM_PEAK_GBS,
                                               "peak_GBs": HBM_PEAK_GBS, "l2_hit_rate": inp.get("l2_hit_rate")},
        "algorithmic_equiv_GBs": algorithmic_bytes / sec_eff / 1e9,
        "algorithmic_bytes_per_launch": algorithmic_bytes,
        "from_profiles_tag": inp.get("from_profiles_tag") if same else None,
        "counts_scaled": scale if same else None,  # scenes per launch here / in the profiled launch (counts are sums over a scene's workgroups)
        "profiled_workload": inp.get("workload") if same else None,
        "calibration_tag": whole.get("calibration_tag") if same else None,
        "note": "bound = VALU instruction issue: peak = 1024 SIMDs x 2.4 GHz / the mean issue cycles of this kernel's own instruction mix "
                "(SQ_INSTS_VALU_* classes priced by tools/valu_peak.hip: 2 cycles full rate, 4 for f64 / conversions / int64, 8 transcendental; "
                "uncategorised and int32 at 2: a lower bound, frac errs low); counts per launch from profiles/<from_profiles_tag>_pmc_*.csv via "
                "tools/roofline.py, durations measured in this run; algorithmic_equiv_GBs (SURVEY 8d: 32 + 128 O bytes per point) is not a "
                "fraction of anything: the kernel proves most pairs zero in registers" + ("" if same else "; profiles/roofline_inputs.json holds no workload with this shape: counts omitted"),
    }
    return block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--bench", default=None, help="a bench.py JSON line to recompute (default: profiles/<tag>_bench.json)")
    ap.add_argument("--no-write", action="store_true")
    ap.add_argument("--other", action="store_true", help="add the tag's workload beside the primary one (another shape bench.py times: rank share, config 5, 128 goals)")
    args = ap.parse_args()
    prof = ROOT / "profiles"
    inp = derive_inputs(args.tag, prof)
    import tempfile
    inputs_path = prof / "roofline_inputs.json"
    if args.no_write:  # recompute an older tag's bench line from that tag's own CSVs without touching the tracked inputs
        inputs_path = Path(tempfile.mkdtemp()) / "roofline_inputs.json"
        inputs_path.write_text(json.dumps(inp, indent=1) + "\n")
    else:
        # the tracked file holds ONE primary workload (the bench default: top level) and any number of others (inp["others"]): a tag
        # replaces the entry with its own workload, wherever that is
        cur = json.loads(inputs_path.read_text()) if inputs_path.exists() else {}
        others = [e for e in (cur.get("others") or []) if e.get("workload") != inp.get("workload")]
        top = {k: v for k, v in cur.items() if k != "others"}
        if args.other:
            if top.get("workload") == inp.get("workload"):
                raise SystemExit("this workload is the primary entry: collect it without --other")
            others.append(inp)
            top["others"] = others
            inputs_path.write_text(json.dumps(top, indent=1) + "\n")
        else:
            inp2 = dict(inp)
            inp2["others"] = others
            inputs_path.write_text(json.dumps(inp2, indent=1) + "\n")
    print(json.dumps(inp, indent=1))
    bench = Path(args.bench) if args.bench else prof / f"{args.tag}_bench.json"
    if bench.exists():
        line = [l for l in bench.read_text().splitlines() if l.startswith("{")][-1]
        b = json.loads(line)
        r = b["roofline"]
        mine = roofline_block(inputs_path, r["avg_launch_ms"], r["launches"], r["timing_stride"], r["algorithmic_bytes_per_launch"],
                              inp["workload"] or {}, r.get("launches_per_step", 1), r.get("ms_per_step"))
        ok = True
        for k in ("achieved", "peak", "frac", "useful_frac", "traffic", "algorithmic_equiv_GBs"):
            a_, b_ = mine.get(k), r.get(k)
            same = (a_ is None and b_ is None) or (a_ is not None and b_ is not None and abs(a_ - b_) <= 5e-4 * max(abs(a_), abs(b_)))
            ok &= same
            print(f"{k:24s} recomputed {a_!r:24} bench line {b_!r:24} {'ok' if same else 'DIFFERENT'}")
        if mine.get("pairs") != r.get("pairs"):
            ok = False
            print("pairs DIFFERENT", mine.get("pairs"), r.get("pairs"))
        if inp["avg_launch_ns_rocprof_kernel_trace"]:
            print(f"avg launch: bench.py HIP events {r['avg_launch_ms'] * 1e3:.1f} us, rocprofv3 --kernel-trace --stats {inp['avg_launch_ns_rocprof_kernel_trace'] / 1e3:.1f} us")
        print("roofline frac", mine["frac"], "<= 1:", mine["frac"] is not None and mine["frac"] <= 1.0)
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
