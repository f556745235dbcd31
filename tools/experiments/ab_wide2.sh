#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local ms=$(env "${envs[@]}" python3 bench.py "$@" --no-plan --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],5), d['config']['layout'])")
  echo "$label | $* | $ms"
}
for shape in "--scenes 50 --goals 64 --waypoints 64" "--scenes 100 --goals 64 --waypoints 64" "--scenes 8 --goals 64 --waypoints 64" "--scenes 50 --goals 64 --waypoints 56" "--scenes 50 --goals 64 --waypoints 52"; do
  run "w4" OMGX_GS_WIDE=0 -- $shape
  run "w6" OMGX_GS_WIDE6_LONG_MAX=100000 -- $shape
  run "w8" OMGX_GS_WIDE6_LONG_MAX=100000 OMGX_GS_WIDE_LONG_W=8 -- $shape
done
for shape in "--scenes 10 --goals 64" "--scenes 12 --goals 64" "--scenes 18 --goals 64" "--scenes 20 --goals 64" "--scenes 6 --goals 128" "--scenes 9 --goals 128"; do
  run "rule w4" OMGX_GS_WIDE=0 -- $shape
  run "rule w8<=384" OMGX_GS_WIDE8_MAX=384 -- $shape
  run "rule w8<=448" OMGX_GS_WIDE8_MAX=448 -- $shape
  run "whole p3 w8<=448" OMGX_GS_WIDE8_MAX=448 -- $shape --goal-parts 1 --pipeline 3
  run "whole p1 w8<=1200" OMGX_GS_WIDE8_MAX=1200 -- $shape --goal-parts 1 --pipeline 1
done
