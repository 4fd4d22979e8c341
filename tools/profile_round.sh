#!/bin/bash
# Round-end evidence on the GPU box: the default bench, a rocprofv3 kernel-trace/stats run
# of the same command, and the two PMC traffic passes.  Outputs under gpurun_out/.
# usage: tools/profile_round.sh <tag>
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R && timeout -k 10 300 python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --no-cpu-baseline --no-alt-mode > $R/gpurun_out/prof_bench.log 2>&1 || exit 1
cp $(find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kernel_stats_$TAG.csv
cd $R && timeout -k 10 500 bash tools/traffic_run.sh || exit 1
cp gpurun_out/traffic.json gpurun_out/traffic_$TAG.json
tail -c 600 gpurun_out/bench_$TAG.json
