#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for sh in "16 64 50 12" "100 64 50 4" "32 64 50 12" "8 64 50 12"; do
for t in "1,5,0,0" "1,10,0,0" "1,10,25,0" "1,10,13,0" "1,10,8,0" "1,5,13,0"; do
  r=$(OMGX_LAYER_ONLY_TILING=$t python3 tools/experiments/plan_once_n.py $sh 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
  echo "$sh | layer-only tiling $t | plan ms $r"
done; done
