#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local ms=$(env "${envs[@]}" python3 bench.py "$@" --no-plan --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],5), d['config']['layout'], d['parity_sample']['ok'], d['parity_sample']['max_cost_rel_err'])")
  echo "$label | $* | $ms"
}
for shape in "--scenes 16 --goals 64 --waypoints 50 --objects 12" "--scenes 8 --goals 64 --waypoints 50 --objects 12" "--scenes 32 --goals 64 --waypoints 50 --objects 12" "--scenes 100 --goals 64 --waypoints 50" "--scenes 50 --goals 64 --waypoints 64" "--scenes 100 --goals 64 --waypoints 41"; do
  run "rule" OMGX_GS_RANGE_MIN=1000000 -- $shape
  for p in 1 2 3; do
    run "tiles (2,$p)" OMGX_GS_RANGE_MIN=1000000 -- $shape --goal-parts 2 --pipeline $p
    run "ranges (2,$p)" OMGX_GS_RANGE_MIN=40 -- $shape --goal-parts 2 --pipeline $p
  done
done
