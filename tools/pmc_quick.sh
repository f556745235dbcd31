#!/bin/bash
# quick PMC pass over the A/B tool: bash tools/pmc_quick.sh <tag> "<counters>" [env assignments...]
TAG=$1; CNT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=/tmp/pmcq_$TAG
mkdir -p $R/gpurun_out $T
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc $CNT --output-format csv -d $T -o $TAG -- python3 $R/tools/ab_goalset.py --sched none --iters 5 > $R/gpurun_out/${TAG}_pmcq.log 2>&1
python3 $R/tools/pmc_summary.py $T $R/gpurun_out/${TAG}_pmcq.csv
grep goalset $R/gpurun_out/${TAG}_pmcq.csv
