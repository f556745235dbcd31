#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# The default iteration is two launches on one stream, so nothing co-runs with the profiled kernel (the device-wide PMC
# counters are attributed cleanly).
# Writes gpurun_out/<tag>_kernel_stats.csv, <tag>_pmc_<COUNTER>.csv and <tag>_traffic.json; copy the
# ones to be judged into profiles/.  PMC passes are separate runs (no trace domains beside kernel-trace).
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=/tmp/prof_$TAG
mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats -o $TAG -- python3 $R/bench.py --no-cpu-baseline --no-plan > $O/${TAG}_bench_under_rocprof.log 2>&1
cp $T/stats/*kernel_stats*.csv $O/${TAG}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --output-format csv -d $T/pmc_$n -o $TAG -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plan > $O/${TAG}_pmc_$n.log 2>&1
  python3 $R/tools/pmc_summary.py $T/pmc_$n $O/${TAG}_pmc_$n.csv
done
python3 - $O $TAG <<'PY'
import csv, json, sys
o, tag = sys.argv[1], sys.argv[2]
def mean(counter, kernel_sub):
    for r in csv.DictReader(open(f"{o}/{tag}_pmc_{counter}.csv")):
        if kernel_sub in r["kernel"] and r["counter"] == counter:
            return float(r["mean_per_dispatch"])
    return None
k = "k_goalset_compact"  # dominant kernel: the goal-set batch
if mean("FETCH_SIZE", k) is None:
    k = "k_sdf_chunks<false"
f, w = mean("FETCH_SIZE", k), mean("WRITE_SIZE", k)
def mean2(fname, counter):
    for r in csv.DictReader(open(f"{o}/{tag}_pmc_{fname}.csv")):
        if k in r["kernel"] and r["counter"] == counter:
            return float(r["mean_per_dispatch"])
    return None
valu_q, gui = mean2("SQ_WAVES", "SQ_ACTIVE_INST_VALU"), mean2("GRBM_GUI_ACTIVE", "GRBM_GUI_ACTIVE")
hit, miss = mean2("TCC_HIT_sum", "TCC_HIT_sum"), mean2("TCC_HIT_sum", "TCC_MISS_sum")
# rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB; MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports
# half the bytes of wide coalesced reads -> doubled as prescribed (an upper bound for this gather pattern).
out = {"kernel": k + " (goal-set batch)", "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
       "goalset_kernel_bytes_per_launch": None if f is None else (2 * f + (w or 0)) * 1024,
       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md gfx950 correction; WRITE_SIZE as reported",
       # SQ_ACTIVE_INST_VALU counts quad-cycles summed over waves; 1024 SIMDs; GRBM_GUI_ACTIVE is summed over 8 XCDs
       "valu_busy_frac": None if not (valu_q and gui) else 4.0 * valu_q / (1024.0 * gui / 8.0),
       "l2_hit_rate": None if not (hit and miss) else hit / (hit + miss)}
json.dump(out, open(f"{o}/{tag}_traffic.json", "w"), indent=1)
print(json.dumps(out))
PY
head -6 $O/${TAG}_kernel_stats.csv | cut -c1-150
