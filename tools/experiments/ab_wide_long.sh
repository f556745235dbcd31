#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local ms=$(env "${envs[@]}" python3 bench.py "$@" --no-plan --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],5), d['config']['layout'])")
  echo "$label | $* | $ms"
}
for shape in "--scenes 16 --goals 64 --waypoints 50 --objects 12" "--scenes 8 --goals 64 --waypoints 50 --objects 12" "--scenes 32 --goals 64 --waypoints 50 --objects 12" "--scenes 100 --goals 64 --waypoints 50" "--scenes 50 --goals 64 --waypoints 64"; do
  run "rule w4" OMGX_GS_WIDE=0 -- $shape
  for p in 1 2 3; do
    run "whole w4 p$p" OMGX_GS_WIDE=0 -- $shape --goal-parts 1 --pipeline $p
    run "whole w6 p$p" OMGX_GS_WIDE6_LONG_MAX=100000 -- $shape --goal-parts 1 --pipeline $p
  done
done
