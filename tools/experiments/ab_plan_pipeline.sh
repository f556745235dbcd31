#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for sh in "13 128 30 4" "16 64 30 4" "25 64 30 4" "50 64 30 4" "100 64 30 4"; do
for p in 1 2 3; do
  r=$(OMGX_PLAN_PIPELINE=$p python3 tools/experiments/plan_once_n.py $sh 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
  e=$(OMGX_PLAN_PIPELINE=$p python3 tools/experiments/plan_once_n.py $sh 1 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
  echo "$sh | pipeline $p | plan ms $r early-stop $e"
done; done
