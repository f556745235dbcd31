#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r02h'
# Writes into gpurun_out/ (copy what is to be judged into profiles/, then `python tools/roofline.py --tag <tag>`):
#   <tag>_bench.json            the unprofiled default bench.py line (run LAST: its roofline block uses this collection's counts)
#   <tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command
#   <tag>_pmc_<SET>.csv         per-kernel means of every counter set (separate --pmc passes, kernel-trace only; MEM_* = TA / TCP / TD)
#   <tag>_workload.json         the workload the counts belong to
#   <tag>_pmc_MIX{1,2}.csv      the kernel's dynamic VALU mix (SQ_INSTS_VALU_* classes: what tools/roofline.py prices)
#   <tag>_pmc_NOEXACT.csv       SQ_INSTS_VALU of the build without the exact path (libomg_hip_noexact.so) -> <tag>_useful.json
#   <tag>_block_counts.json     tools/gs_block_counts.py --json (libomg_hip_cnt.so): block counts per goal workgroup, pair statistics
#   <tag>_valu_peak.{txt,csv}   tools/valu_peak.hip: cycles per wave64 VALU instruction per SIMD (calibration of the bound), with
#   <tag>_valu_peak_pmc.csv     the same kernels under GRBM_GUI_ACTIVE / SQ_INSTS_VALU (set VALU_PEAK=1: the table only changes with the hardware)
TAG=${1:-r02}
# BENCH_ARGS: the shape to profile, e.g. "--goals 128", "--scenes 13 --goals 128", "--scenes 16 --waypoints 50 --objects 12" (default: bench.py's own)
# OTHER=1: file the counts beside the primary workload of profiles/roofline_inputs.json (tools/roofline.py --other)
BENCH_ARGS=${BENCH_ARGS:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
T=/tmp/prof_$TAG
mkdir -p $O $T
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $T/stats -o $TAG -- python3 $R/bench.py $BENCH_ARGS --no-cpu-baseline --no-plan --no-parity > $O/${TAG}_bench_under_rocprof.log 2>&1
cp $T/stats/*kernel_stats*.csv $O/${TAG}_kernel_stats.csv
run_pmc() {  # name, counters
  rocprofv3 --pmc $2 --output-format csv -d $T/pmc_$1 -o $TAG -- python3 $R/bench.py $BENCH_ARGS --steps 5 --warmup 1 --no-cpu-baseline --no-plan --no-parity > $O/${TAG}_pmc_$1.log 2>&1
  python3 $R/tools/pmc_summary.py $T/pmc_$1 $O/${TAG}_pmc_$1.csv
}
run_pmc FETCH_SIZE "FETCH_SIZE"
run_pmc WRITE_SIZE "WRITE_SIZE"
run_pmc TCC "TCC_HIT_sum TCC_MISS_sum"
run_pmc SQ_INSTS "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
run_pmc LANES "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
run_pmc SQ_WAIT "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
run_pmc GRBM "GRBM_GUI_ACTIVE"
# the vector-memory path of the gathers: address unit, L1, data return (how busy each is next to the VALU)
if [ -z "$SKIP_MEMPIPE" ]; then
run_pmc MEM_TA "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TA_FLAT_READ_WAVEFRONTS_sum"
run_pmc MEM_TCP "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
run_pmc MEM_TD "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
fi
python3 - $O/${TAG}_bench_under_rocprof.log > $O/${TAG}_workload.json <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
c = json.loads(line)["config"]
print(json.dumps({"scenes": c["scenes_per_gpu"], "goals": c["goals"], "waypoints": c["waypoints"], "points_per_link": 15, "grid": 64,
                  "pipeline": c["pipeline_parts"], "objects": c["objects_per_scene"]}))
PY
run_pmc MIX1 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
run_pmc MIX2 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
if [ -f $R/omg-planner_amd/csrc/libomg_hip_noexact.so ]; then
  rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $T/pmc_NOEXACT -o $TAG -- python3 $R/tools/bench_variant.py libomg_hip_noexact.so $BENCH_ARGS --steps 5 --warmup 1 --no-cpu-baseline --no-plan --no-parity > $O/${TAG}_pmc_NOEXACT.log 2>&1
  python3 $R/tools/pmc_summary.py $T/pmc_NOEXACT $O/${TAG}_pmc_NOEXACT.csv
fi
if [ -f $R/omg-planner_amd/csrc/libomg_hip_cnt.so ] && [ -z "$BENCH_ARGS" ]; then  # the block counts belong to the default shape
  (cd $R && python3 tools/gs_block_counts.py 100 --json $O/${TAG}_block_counts.json > $O/${TAG}_block_counts.txt 2>&1)
fi
python3 $R/tools/make_useful.py $TAG $O > /dev/null 2>&1
if [ -n "$VALU_PEAK" ] && [ -x $R/tools/_build/valu_peak ]; then
  $R/tools/_build/valu_peak $O/${TAG}_valu_peak.csv > $O/${TAG}_valu_peak.txt 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $T/vp -o vp -- $R/tools/_build/valu_peak pmc > $O/${TAG}_valu_peak_pmc.txt 2>&1
  find $T/vp -name "*counter_collection*.csv" -exec cp {} $O/${TAG}_valu_peak_pmc_raw.csv \;
fi
# the unprofiled bench line LAST, with the per-launch counts of THIS collection behind its roofline block
cp $O/${TAG}_*.csv $O/${TAG}_*.json $R/profiles/ 2>/dev/null
(cd $R && python3 tools/roofline.py --tag $TAG ${OTHER:+--other} > $O/${TAG}_roofline_inputs.log 2>&1; cp profiles/roofline_inputs.json $O/${TAG}_roofline_inputs.json)
if [ -z "$BENCH_ARGS" ]; then python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log
else python3 $R/bench.py $BENCH_ARGS --no-plan --no-cpu-baseline > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log; fi
head -4 $O/${TAG}_kernel_stats.csv | cut -c1-160
grep goalset $O/${TAG}_pmc_*.csv | cut -c1-200
cut -c1-1200 $O/${TAG}_bench.json
