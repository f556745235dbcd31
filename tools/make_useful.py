#!/usr/bin/env python3
"""profiles/<tag>_useful.json: how much of what the dominant kernel issues is exact-path work, and what its triage lets through.

    python tools/make_useful.py <tag> [dir]     (dir: where <tag>_pmc_SQ_INSTS.csv, <tag>_pmc_NOEXACT.csv, <tag>_block_counts.json lie)

exact_path_valu_share = (SQ_INSTS_VALU of the shipped kernel - SQ_INSTS_VALU of the same kernel built with -DOMGX_GS_NO_EXACT,
the exact path compiled out) / SQ_INSTS_VALU of the shipped kernel, both per launch on the bench workload (separate --pmc passes
of tools/collect_profiles.sh).  pairs: tools/gs_block_counts.py --json (counting build)."""
import csv
import json
import sys
from pathlib import Path

tag = sys.argv[1]
d = Path(sys.argv[2]) if len(sys.argv) > 2 else Path(__file__).resolve().parents[1] / "profiles"


def valu(name):
    for r in csv.DictReader(open(d / f"{tag}_pmc_{name}.csv")):
        if any(k in r["kernel"] for k in ("k_goalset_queue<2, false, false, false, false, 4>", "k_goalset_queue<2, false, false, false, false, 8>", "k_goalset_queue<2, false, false, true, false, 4>", "k_goalset_queue<2, false, false, false, false>", "k_goalset_queue<2, false, false, true, false>", "k_goalset_queue<2, false, false, false>", "k_goalset_queue<2, false, false>")) and r["counter"] == "SQ_INSTS_VALU":
            return float(r["mean_per_dispatch"])
    return None


full, triage = valu("SQ_INSTS"), valu("NOEXACT")
bc = json.loads((d / f"{tag}_block_counts.json").read_text()) if (d / f"{tag}_block_counts.json").exists() else {}
out = {"exact_path_valu_share": None if not (full and triage) else (full - triage) / full,
       "valu_wave_insts_per_launch": full, "valu_wave_insts_per_launch_without_exact_path": triage,
       "pairs": bc.get("pairs"), "per_goal_workgroup": bc.get("per_goal_workgroup")}
(d / f"{tag}_useful.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
