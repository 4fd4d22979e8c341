#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 2 --warmup 1 --blocks 4 --no-cpu-baseline --no-profile ${@} > $R/gpurun_out/prof_bench.log 2>&1
tail -2 $R/gpurun_out/prof_bench.log | cut -c1-300
