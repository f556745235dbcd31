#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for sh in "16 64 50 12" "8 64 50 12" "13 128 30 4" "16 64 30 4" "4 64 30 4" "2 64 30 4"; do
for v in 0 32; do
  r=$(OMGX_SMOOTH_SINGLE_PART_BELOW=$v python3 tools/experiments/plan_once_n.py $sh 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
  echo "$sh | single part below $v scenes | plan ms $r"
done; done
