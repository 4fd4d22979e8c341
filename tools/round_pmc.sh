#!/bin/bash
# SQ counters of the final kernels on the two 70B MLP shapes (+ a single-stream rocprofv3 stats run of the bench,
# whose per-kernel averages are the ones roofline.avg_launch_ms has to agree with).  usage: tools/round_pmc.sh <tag>
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
cd $R
for s in "28672 8192 3" "8192 28672 3"; do
  set -- $s
  rm -rf gpurun_out/prof_trace gpurun_out/prof_pmcA gpurun_out/prof_pmcB
  timeout -k 10 400 bash tools/pmc_run.sh $1 $2 $3 > /dev/null 2>&1 || exit 1
  python3 tools/pmc_summary.py gpurun_out/prof_pmcA gpurun_out/prof_pmcB > gpurun_out/${TAG}_pmc_sq_counters_$1x$2_k$3.txt
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench1 -- python3 $R/bench.py --streams 1 --no-cpu-baseline --no-alt-mode --no-live-traffic > $R/gpurun_out/prof_bench1.log 2>&1 || exit 1
cp $(find $R/gpurun_out/prof_bench1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_rocprofv3_kernel_stats_bench_streams1.csv
tail -1 $R/gpurun_out/prof_bench1.log | cut -c1-400
