#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o pt -- python3 $R/tools/experiments/plan_once.py "$@" > $R/gpurun_out/plan_once.log 2>&1
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1)
python3 $R/tools/experiments/plan_periods.py $f
