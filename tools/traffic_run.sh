#!/bin/bash
# HBM traffic of every kernel on the bench workload: two separate PMC passes (FETCH_SIZE,
# WRITE_SIZE), kernel-trace/stats only (no other trace domains), program directly after `--`.
#   tools/traffic_run.sh [workload_id [k [blocks]]]     (default: llama3-70b-slice 3 1)
# Writes gpurun_out/traffic.json stamped with the workload; copy it to profiles/traffic_latest.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-llama3-70b-slice}
K=${2:-3}
BLOCKS=${3:-1}
ARGS="--workload $WL --k $K --blocks $BLOCKS --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-profile --no-alt-mode --prewarm-seconds 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_write.log 2>&1 &&
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/traffic.json $WL $K
