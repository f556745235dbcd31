#!/bin/bash
# The step on the other shapes DESIGN.md section 5.3 quotes (GPU box): one line per shape — ms per step, scene-iterations/s, the layout
# ChompEngine.layout chose, parity sample ok.   gpurun -- 'bash tools/sensitivity.sh > gpurun_out/<tag>_sensitivity.txt'
cd ${GRAFT_REPO_ROOT:-$(pwd)}
while IFS= read -r a; do
  python3 bench.py $a --steps 100 --no-plan --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sens $a:', round(d['ms_per_step'], 4), round(d['value']), d['config'].get('layout'), d['parity_sample']['ok'])"
done <<'SHAPES'
--goals 128
--waypoints 50
--waypoints 40
--waypoints 64 --scenes 50
--scenes 400
--scenes 200 --goals 16
--share-grids
--ol-alg FTL
--scenes 13 --goals 128
--scenes 25
--scenes 50
--scenes 16 --waypoints 50 --objects 12
--scenes 16
--scenes 8
--scenes 4
--scenes 2
--scenes 1
SHAPES
