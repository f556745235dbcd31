#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for t in "1,5,0,0" "1,10,0,0" "1,5,15,0" "1,10,15,0" "1,10,8,0" "1,5,8,0"; do
  for early in 0 1; do
    r=$(OMGX_LAYER_ONLY_TILING=$t python3 tools/experiments/plan_once.py 100 64 $early 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
    echo "tiling $t early=$early best plan ms $r"
  done
done
for t in "1,5,0,0" "1,10,15,0"; do
  r=$(OMGX_LAYER_ONLY_TILING=$t python3 tools/experiments/plan_once.py 13 128 0 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
  echo "13x128 tiling $t plan ms $r"
  r=$(OMGX_LAYER_ONLY_TILING=$t python3 tools/experiments/plan_once.py 16 64 0 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
  echo "16x64 tiling $t plan ms $r"
done
