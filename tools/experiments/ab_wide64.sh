#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local ms=$(env "${envs[@]}" python3 bench.py "$@" --no-plan --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],5), d['config']['layout'])")
  echo "$label | $* | $ms"
}
for shape in "--scenes 4 --goals 64 --waypoints 64" "--scenes 8 --goals 64 --waypoints 64" "--scenes 16 --goals 64 --waypoints 64" "--scenes 32 --goals 64 --waypoints 64" "--scenes 16 --goals 64 --waypoints 60 --objects 12"; do
  run "rule" X=1 -- $shape
  for p in 1 2 3; do
    run "whole w8 p$p" X=1 -- $shape --goal-parts 1 --pipeline $p
  done
  run "whole w4 p3" OMGX_GS_WIDE=0 -- $shape --goal-parts 1 --pipeline 3
done
