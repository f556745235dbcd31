#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for sh in "16 64 50 12" "8 64 50 12" "32 64 50 12" "13 128 30 4" "16 64 30 4" "100 64 30 4" "100 64 50 4"; do
for t in "default" "1,10,4,1" "1,10,8,1" "1,5,8,1" "1,10,16,1"; do
  if [ "$t" = default ]; then e="X=1"; else e="OMGX_LAYER_ONLY_TILING=$t"; fi
  r=$(env $e python3 tools/experiments/plan_once_n.py $sh 2>/dev/null | grep "plan ms" | awk '{print $3}' | sort -n | head -1)
  echo "$sh | layer-only tiling $t | plan ms $r"
done; done
