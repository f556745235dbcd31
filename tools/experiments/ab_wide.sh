#!/bin/bash
# Wide goal workgroups (six / eight waves) against four on bench shapes: bash tools/experiments/ab_wide.sh  (GPU box)
cd ${GRAFT_REPO_ROOT:-.}
run() {  # label, env..., -- bench args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local ms=$(env "${envs[@]}" python3 bench.py "$@" --no-plan --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],5), d['config']['layout'])")
  echo "$label | $* | $ms"
}
for shape in "--scenes 13 --goals 128" "--scenes 25 --goals 64" "--scenes 16 --goals 64" "--scenes 50 --goals 64"; do
  for rep in 1 2; do
    run "w4" OMGX_GS_WIDE=0 -- $shape
    run "w6" OMGX_GS_WIDE6_MAX=1200 -- $shape
    run "w8" OMGX_GS_WIDE8_MAX=1200 -- $shape
  done
done
for shape in "--scenes 8 --goals 64" "--scenes 4 --goals 64" "--scenes 2 --goals 64"; do
  run "rule w4" OMGX_GS_WIDE=0 -- $shape
  run "whole goals w4 p2" OMGX_GS_WIDE=0 -- $shape --goal-parts 1 --pipeline 2
  run "whole goals w8 p2" OMGX_GS_WIDE8_MAX=1200 -- $shape --goal-parts 1 --pipeline 2
  run "whole goals w8 p1" OMGX_GS_WIDE8_MAX=1200 -- $shape --goal-parts 1 --pipeline 1
  run "whole goals w6 p2" OMGX_GS_WIDE6_MAX=1200 -- $shape --goal-parts 1 --pipeline 2
done
