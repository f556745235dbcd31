#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for sh in "13 128 30 4" "16 64 30 4" "25 64 30 4" "8 64 30 4"; do
for v in 0 32; do
for t in "1,5,0,0" "1,10,0,0" "1,10,8,0"; do
  r=$(OMGX_SMOOTH_SINGLE_PART_BELOW=$v OMGX_LAYER_ONLY_TILING=$t python3 tools/experiments/plan_once_n.py $sh 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
  echo "$sh | single part below $v scenes | tiling $t | plan ms $r"
done; done; done
