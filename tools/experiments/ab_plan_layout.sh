#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for sh in "5 64 30 4" "6 64 30 4" "8 64 30 4" "10 64 30 4" "12 64 30 4" "6 128 30 4" "8 64 50 12" "16 64 50 12"; do
  echo "$sh | rule | $(python3 tools/experiments/plan_once_n.py $sh 2>/dev/null | grep "plan ms" | awk '{print $3, $4, $5, $6, $7, $8, $9}' | sort -n | head -1)"
  echo "$sh | latency | $(OMGX_PLAN_LATENCY=1 python3 tools/experiments/plan_once_n.py $sh 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)"
  for gp in 1 2; do for p in 1 2 3; do
    echo "$sh | goal_parts $gp pipeline $p | $(OMGX_PLAN_GOAL_PARTS=$gp OMGX_PLAN_PIPELINE=$p python3 tools/experiments/plan_once_n.py $sh 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)"
  done; done
done
