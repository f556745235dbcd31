#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for sh in "13 128 30 4" "16 64 30 4" "4 64 30 4" "2 64 30 4" "8 64 30 4" "25 64 30 4" "6 128 30 4"; do
for rep in 1 2; do
for v in 0 16; do
  r=$(OMGX_GS_LAYER_SMALL=$v python3 tools/experiments/plan_once_n.py $sh 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
  echo "$sh | small-launch layer rule $v | plan ms $r"
done; done; done
