#!/bin/bash
# kernel timeline of a bench shape under rocprofv3 --kernel-trace: bash tools/experiments/timeline_shape.sh <tag> <bench args...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o tl -- python3 $R/bench.py "$@" --steps 60 --warmup 10 --no-cpu-baseline --no-plan --no-parity > $R/gpurun_out/${TAG}_bench.log 2>&1
f=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timeline.py $f --last 48 --skip-tail 40 > $R/gpurun_out/${TAG}_timeline.txt
cat $R/gpurun_out/${TAG}_timeline.txt
