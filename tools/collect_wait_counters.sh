#!/bin/bash
# Attribution of the goal-set kernel's wait time on the GPU box (separate --pmc passes, kernel-trace only):
#   gpurun --timeout 1500 -- 'bash tools/collect_wait_counters.sh r02a'
# Writes gpurun_out/<tag>_pmc_wait{A,B,C,D}.csv (per kernel means, tools/pmc_summary.py), the VALU-rate calibration
# (<tag>_valu_rates.txt, <tag>_valu_rates_pmc.csv) and, when the instrumented library variant was built, <tag>_gs_phase_clock.json.
TAG=${1:-r02a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=/tmp/prof_$TAG
mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
$R/tools/_build/valu_rates > $O/${TAG}_valu_rates.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $T/vr -o vr -- $R/tools/_build/valu_rates pmc > $O/${TAG}_valu_rates_pmc.log 2>&1
python3 - $T/vr $O/${TAG}_valu_rates_pmc.csv <<'PY'
import csv, glob, sys, collections
d, out = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for f in glob.glob(d + "/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Dispatch_Id"], r["Kernel_Name"][:40], r["Grid_Size"])
        rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
w = csv.writer(open(out, "w"))
names = ["SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"]
w.writerow(["dispatch", "kernel", "grid"] + names + ["valu_busy = 4*ACTIVE_INST_VALU/(1024*GUI_ACTIVE/8)"])
for k, v in rows.items():
    busy = 4.0 * v.get("SQ_ACTIVE_INST_VALU", 0) / (1024.0 * v.get("GRBM_GUI_ACTIVE", 1) / 8.0) if v.get("GRBM_GUI_ACTIVE") else ""
    w.writerow(list(k) + [v.get(n, "") for n in names] + [busy])
PY
if [ -f $R/omg-planner_amd/csrc/libomg_hip_clk.so ]; then
  python3 $R/tools/gs_phase_clock.py > $O/${TAG}_gs_phase_clock.json 2> $O/${TAG}_gs_phase_clock.log
fi
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_INSTS_SALU" \
         "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SMEM SQ_BUSY_CYCLES" \
         "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM_RD" \
         "SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE"; do
  n=$(echo ABCDE | cut -c$((i+1)))
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $T/pmc_$n -o $TAG -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-plan > $O/${TAG}_pmc_wait$n.log 2>&1
  python3 $R/tools/pmc_summary.py $T/pmc_$n $O/${TAG}_pmc_wait$n.csv
done
cat $O/${TAG}_pmc_wait*.csv | grep goalset
