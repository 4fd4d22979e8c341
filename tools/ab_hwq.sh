#!/bin/bash
# hardware queues per process (ROCm default 4) against the bench's 8 engines (+ 8 side streams in reference_cpu mode)
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for q in 4 8 16 24; do
  echo -n "GPU_MAX_HW_QUEUES=$q: "
  GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python3 -c '
import json, sys
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("ref %.1f GB/s %.1f ms | exact %.1f GB/s %.1f ms" % (r["value"], r["ms_per_step"], r["alt_norm_mode"]["value"], r["alt_norm_mode"]["ms_per_step"]))'
done
done
