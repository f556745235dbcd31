#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
cat > /tmp/plan_n.py <<'P'
import copy, sys, time
sys.path.insert(0, ".")
import torch, bench
from omg_planner_amd.engine import ChompEngine
S, G, n, obj = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg, model, batch, start, goals = bench.build_workload(S, G, n, 64, 0, False, num_objects=obj)
eng = ChompEngine.auto(model, batch, copy.deepcopy(cfg), start, goals, layout_scenes=S, device=torch.device("cuda:0"), ol_alg="MD")
snap = eng.snapshot()
best = 1e9
for rep in range(6):
    eng.restore(snap); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.plan(early_stop=False); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) * 1e3)
print(round(best, 3), eng.layout_used)
P
for sh in "16 64 50 12" "100 64 50 4" "100 64 41 4" "50 64 64 4" "32 64 50 12"; do
  for f in 0 12 20; do
    echo "layer follows the window (min $f) | $sh: $(OMGX_GS_LAYER_FOLLOW=$f python3 /tmp/plan_n.py $sh 2>/dev/null | tail -1)"
  done
done
