#!/bin/bash
# A/B of a variant library against the shipped one on bench shapes, alternating: bash tools/experiments/ab_variant.sh <variant .so> <runs> -- <bench args...>
V=$1; N=$2; shift 3
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq $N); do
  a=$(python3 bench.py "$@" --no-plan --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  b=$(python3 tools/bench_variant.py $V "$@" --no-plan --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
  echo "shipped $a variant $b"
done
