#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptn
rocprofv3 --kernel-trace --output-format csv -d /tmp/ptn -o pt -- python3 $R/tools/experiments/plan_once_n.py "$@" > $R/gpurun_out/plan_once_n.log 2>&1
f=$(find /tmp/ptn -name "*kernel_trace.csv" | head -1)
python3 $R/tools/experiments/plan_periods.py $f
