#!/bin/bash
# A/B of the norm walker's side stream on the default bench (8 engines): priority default / high / low, and no side stream
R=$GRAFT_REPO_ROOT
cd $R
run() {
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-alt-mode --no-profile 2>/dev/null | python3 -c '
import json, sys
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("%.1f GB/s %.1f ms/step" % (r["value"], r["ms_per_step"]))'
}
for rep in 1 2; do
  echo -n "default priority: "; SMHIP_AUX_PRIORITY=0 run
  echo -n "high priority:    "; SMHIP_AUX_PRIORITY=1 run
  echo -n "low priority:     "; SMHIP_AUX_PRIORITY=2 run
  echo -n "no side stream:   "; SMHIP_DEBUG=aten_overlap=0 run
  echo -n "exact norms:      "; python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-alt-mode --no-profile --norm-mode exact 2>/dev/null | python3 -c '
import json, sys
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("%.1f GB/s %.1f ms/step" % (r["value"], r["ms_per_step"]))'
done
