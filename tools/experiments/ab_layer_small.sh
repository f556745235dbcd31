#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for sh in "13 128 30 4" "16 64 30 4" "25 64 30 4" "8 64 30 4" "16 64 50 12"; do
for v in 0 8 16 30; do
  r=$(OMGX_GS_LAYER_SMALL=$v python3 tools/experiments/plan_once_n.py $sh 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
  echo "$sh | layer pieces of small launches: ten link groups x blocks of $v | plan ms $r"
done; done
for sh in "--scenes 13 --goals 128" "--scenes 16 --goals 64"; do
for v in 0 16 30; do
  ms=$(OMGX_GS_LAYER_SMALL=$v python3 bench.py $sh --no-plan --no-cpu-baseline --no-parity 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],5))")
  echo "pinned step $sh | $v | $ms"
done; done
