#!/bin/bash
# Differential fuzz campaigns on the GPU box, one JSON line per campaign (tool, seed, trials, agreeing trials, the tool's own
# summary line) under gpurun_out/fuzz/ — copy into profiles/fuzz/ so that DESIGN.md section 2's figures can be checked:
#   gpurun --timeout 3000 -- 'bash tools/run_fuzz_campaign.sh r02 300 400 1000 1000 60 7'
# arguments: tag, trials of fuzz_parity / fuzz_sdf / fuzz_chomp / fuzz_learner / fuzz_misc, seed
TAG=${1:-r02}; NP=${2:-200}; NS=${3:-300}; NC=${4:-800}; NL=${5:-800}; NM=${6:-40}; SEED=${7:-7}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/fuzz
mkdir -p $O
run() {  # tool trials
  local log=$O/${TAG}_$1_seed${SEED}.log
  local t0=$(date +%s)
  timeout 1500 python3 $R/tests/fuzz/$1.py $2 $SEED > $log 2>&1
  local rc=$?
  python3 - "$1" "$2" "$SEED" "$rc" "$log" "$(( $(date +%s) - t0 ))" > $O/${TAG}_$1_seed${SEED}.json <<'PY'
import json, re, sys
tool, trials, seed, rc, log, secs = sys.argv[1:7]
lines = [l.rstrip() for l in open(log, errors="replace") if l.strip()]
summary = next((l for l in reversed(lines) if "trials agree" in l), lines[-1] if lines else "")
m = re.search(r"(\d+)/(\d+) trials agree", summary)
fails = [l for l in lines if "FAIL" in l]
classes = {}
for l in lines:  # "class <name>: <n> trial(s) <json>": trials that agree under a class of their own (diverged in both, free-running feedback)
    mc = re.match(r"class (\w+): (\d+) trial\(s\) (.*)$", l)
    if mc:
        classes[mc.group(1)] = {"trials": int(mc.group(2)), "first": json.loads(mc.group(3))}
print(json.dumps({"tool": "tests/fuzz/" + tool + ".py", "seed": int(seed), "trials": int(trials), "agree": int(m.group(1)) if m else None,
                  "failures": len(fails), "first_failures": fails[:3], "classes": classes, "summary": summary, "exit_code": int(rc), "seconds": int(secs)}))
PY
  cat $O/${TAG}_$1_seed${SEED}.json
}
run fuzz_parity $NP
run fuzz_sdf $NS
run fuzz_chomp $NC
run fuzz_learner $NL
run fuzz_misc $NM
