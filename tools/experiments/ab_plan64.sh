#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
run() {
  local label=$1; shift
  local out=$(python3 bench.py "$@" --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],5), 'plan', round(d['ms_per_plan'],3), 'early', round(d['ms_per_plan_early_stop'],3), d['config']['layout'])")
  echo "$label | $* | $out"
}
for shape in "--scenes 16 --goals 64 --waypoints 64" "--scenes 8 --goals 64 --waypoints 64" "--scenes 16 --goals 64 --waypoints 50 --objects 12" "--scenes 16 --goals 64 --waypoints 60 --objects 12"; do
  run "rule" $shape
  run "whole p2" $shape --goal-parts 1 --pipeline 2
  run "whole p1" $shape --goal-parts 1 --pipeline 1
  run "whole p3" $shape --goal-parts 1 --pipeline 3
done
