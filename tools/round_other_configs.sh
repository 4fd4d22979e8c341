#!/bin/bash
# the other BASELINE configurations and the north-star micro-benchmark, both norm modes in each line (alt_norm_mode)
R=$GRAFT_REPO_ROOT
cd $R
for w in "8192sq --k 2 --blocks 8" "8192sq --k 3 --blocks 8" "llama3-8b --k 2 --blocks 8" "llama3-8b --k 4 --blocks 8"; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python3 -c '
import json, sys
r = json.loads(sys.stdin.read())
a = r["alt_norm_mode"]
print(r["config"]["workload_id"], "K=%d" % r["config"]["k"], "| reference_cpu: %.1f GB/s %.2f ms/step canonical frac %.3f moved frac %.3f | exact: %.1f GB/s %.2f ms/step canonical frac %.3f | cull speculation hit rate %.3f" % (r["value"], r["ms_per_step"], r["pipeline_hbm_frac"], r["pipeline_hbm_frac_moved"], a["value"], a["ms_per_step"], a["pipeline_hbm_frac"], r["cull_speculation"]["hit_rate"]))
'
done
