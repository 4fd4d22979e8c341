cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace2 -- python3 $R/tools/kprof.py 8192 8192 2 2 > $R/gpurun_out/prof_trace2.log 2>&1
